# A/B of library variants through bench.py in one gpurun call: bash tools/ab_lib.sh <tag> "<bench args>" <repeats> <variant|main> [...]
. "$(dirname "$0")/common.sh"
set +e
TAG="${1:?tag}"; ARGS="$2"; N="${3:-2}"; shift 3
O="$R/gpurun_out/$TAG"; mkdir -p "$O"
for r in $(seq 1 "$N"); do
  for v in "$@"; do
    if [ "$v" = main ]; then unset ZYG_AMD_LIB; else export ZYG_AMD_LIB=$R/variants/lib_$v.so; fi
    timeout 600 python3 bench.py --no-cpu-baseline --no-extra $ARGS > "$O/${v}_$r.json" 2> "$O/${v}_$r.err"
    python3 -c "
import json
d=json.load(open('$O/${v}_$r.json')); k=d['roofline']['kernel_ms']; print('%-12s %8.1f Msamples/s  trace %.1f shade %.1f scatter %.1f shadow %.1f  copy %s' % ('$v', d['value'], k['wf_intersect'], k['wf_shade'], k['wf_scatter'], k['wf_shadow'], d['roofline']['hbm']['measured_copy_gbs']))" || tail -3 "$O/${v}_$r.err"
  done
done
