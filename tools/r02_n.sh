cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/r02n
rm -rf $O; mkdir -p $O
for cfg in "outdoor:--scene outdoor" "glass:--scene cornell_glass" "subst:--material substitute" "staged:--nee staged" "persist:--persistent 1" "tick:--fuse-tick 1" "precise:--precise 1" "mesh_precise:--scene cornell_mesh --precise 1" "unfused:--fuse 0"; do
  tag=${cfg%%:*}; args=${cfg#*:}
  timeout 300 python3 bench.py --steps 4 --no-cpu-baseline --no-extra $args > $O/$tag.json 2> $O/$tag.err
  python3 -c "
import json
d=json.load(open('$O/$tag.json')); k=d['roofline']['kernel_ms']; print('%-13s %8.1f Msamples/s  rms %s' % ('$tag', d['value'], d.get('rms_vs_oracle',{}).get('rms')))" || tail -3 $O/$tag.err
done
