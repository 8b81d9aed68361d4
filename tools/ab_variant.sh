# A/B of the main library against variant libraries (zyg_amd/build.py --variant) through bench.py in one gpurun call (same box):
# usage: bash tools/ab_variant.sh "<variant> [<variant> ...]" [repeats] [more bench args]
. "$(dirname "$0")/common.sh"
set +e
VARIANTS="${1:?variants}"; N="${2:-2}"; shift 2 || shift $#
O="$R/gpurun_out/abv"; mkdir -p "$O"
for r in $(seq 1 "$N"); do
  for v in main $VARIANTS; do
    if [ "$v" = main ]; then unset ZYG_AMD_LIB; else export ZYG_AMD_LIB=$R/variants/lib_$v.so; fi
    timeout 300 python3 bench.py --steps 8 --no-cpu-baseline --no-extra "$@" > "$O/$v.json" 2> "$O/$v.err"
    python3 -c "
import json
d=json.load(open('$O/$v.json')); k=d['roofline']['kernel_ms']; print('%-8s %8.1f  trace %.1f shade %.1f scatter %.1f shadow %.1f  rms %s' % ('$v', d['value'], k['wf_intersect'], k['wf_shade'], k['wf_scatter'], k['wf_shadow'], (d.get('rms_vs_oracle') or {}).get('rms')))" || tail -3 "$O/$v.err"
  done
done
