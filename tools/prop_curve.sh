# usage: bash tools/prop_curve.sh: configs[1] with 7 + n solid props (plaques on the back wall) for a few n: the prop tables' cost per prop
. "$(dirname "$0")/common.sh"
set +e
O="$R/gpurun_out/prop_curve"; mkdir -p "$O"
for n in 9 25 57 89 105 106; do
  timeout 300 python3 bench.py --steps 8 --no-cpu-baseline --no-extra --extra-props $n > "$O/p$n.json" 2> "$O/p$n.err"
  python3 -c "
import json,sys
d=json.load(open(sys.argv[1])); k=d['roofline']['kernel_ms']; print('%2d solid props %8.1f Msamples/s  trace %.1f shade %.1f scatter %.1f  rays/sample %s' % (7+int(sys.argv[2]), d['value'], k['wf_intersect'], k['wf_shade'], k['wf_scatter'], d['rays_per_sample']))" "$O/p$n.json" "$n" || tail -3 "$O/p$n.err"
done
