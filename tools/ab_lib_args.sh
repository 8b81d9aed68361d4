# A/B of (library variant, bench arguments) pairs in one gpurun call: bash tools/ab_lib_args.sh <tag> <repeats> "<variant|main>:<bench args>" [...]
. "$(dirname "$0")/common.sh"
set +e
TAG="${1:?tag}"; N="${2:-1}"; shift 2
O="$R/gpurun_out/$TAG"; mkdir -p "$O"
for r in $(seq 1 "$N"); do
  k=0
  for pair in "$@"; do
    k=$((k+1)); v="${pair%%:*}"; a="${pair#*:}"
    if [ "$v" = main ]; then unset ZYG_AMD_LIB; else export ZYG_AMD_LIB=$R/variants/lib_$v.so; fi
    timeout 900 python3 bench.py --no-cpu-baseline --no-extra $a > "$O/${k}_$r.json" 2> "$O/${k}_$r.err"
    python3 -c "
import json
d=json.load(open('$O/${k}_$r.json')); q=d['roofline']['kernel_ms']; print('%-8s %-64s %8.1f Msamples/s  intersect %.1f shade %.1f scatter %.1f shadow %.1f' % ('$v', '''$a''', d['value'], q['wf_intersect'], q['wf_shade'], q['wf_scatter'], q['wf_shadow']))" || tail -3 "$O/${k}_$r.err"
  done
done
