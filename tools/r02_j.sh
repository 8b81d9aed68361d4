cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02j
rm -rf $O; mkdir -p $O
cd $R
for cfg in "mesh:--precise 0 --scene cornell_mesh" "mesh_staged:--precise 0 --scene cornell_mesh --nee staged" "subst:--precise 0 --material substitute" "subst_staged:--precise 0 --material substitute --nee staged" "relaxed_staged:--precise 0 --nee staged"; do
  tag=${cfg%%:*}; args=${cfg#*:}
  timeout 300 python3 bench.py --steps 8 --no-cpu-baseline --no-extra $args > $O/$tag.json 2> $O/$tag.err
  python3 -c "
import json
d=json.load(open('$O/$tag.json')); k=d['roofline']['kernel_ms']; print('%-16s %8.1f ' % ('$tag', d['value']), k)" || tail -3 $O/$tag.err
done
