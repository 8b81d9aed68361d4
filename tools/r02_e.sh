cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02e
rm -rf $O; mkdir -p $O
cd $R
timeout 300 python3 tools/persistent_check.py 2>&1 | tail -8
for cfg in "relaxed:--precise 0" "precise:--precise 1" "mesh:--precise 0 --scene cornell_mesh" "relaxed_p:--precise 0 --persistent 1" "precise_p:--precise 1 --persistent 1" "mesh_p:--precise 0 --scene cornell_mesh --persistent 1"; do
  tag=${cfg%%:*}; args=${cfg#*:}
  timeout 300 python3 bench.py --steps 8 --no-cpu-baseline --no-extra $args > $O/$tag.json 2> $O/$tag.err
  python3 -c "
import json
d=json.load(open('$O/$tag.json')); k=d['roofline']['kernel_ms']; print('%-10s %8.1f  trace %.1f shade %.1f scatter %.1f' % ('$tag', d['value'], k['wf_intersect'], k['wf_shade'], k['wf_scatter']))" || tail -3 $O/$tag.err
done
