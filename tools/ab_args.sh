# A/B of bench.py argument sets in one gpurun call: bash tools/ab_args.sh <tag> <repeats> "<args A>" "<args B>" [...]
. "$(dirname "$0")/common.sh"
set +e
TAG="${1:?tag}"; N="${2:-1}"; shift 2
O="$R/gpurun_out/$TAG"; mkdir -p "$O"
for r in $(seq 1 "$N"); do
  k=0
  for a in "$@"; do
    k=$((k+1))
    timeout 900 python3 bench.py --no-cpu-baseline --no-extra $a > "$O/${k}_$r.json" 2> "$O/${k}_$r.err"
    python3 -c "
import json
d=json.load(open('$O/${k}_$r.json')); q=d['roofline']['kernel_ms']; print('%-70s %8.1f Msamples/s  intersect %.1f shade %.1f scatter %.1f shadow %.1f' % ('''$a''', d['value'], q['wf_intersect'], q['wf_shade'], q['wf_scatter'], q['wf_shadow']))" || tail -3 "$O/${k}_$r.err"
  done
done
