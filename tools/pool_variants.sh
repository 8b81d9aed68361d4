# A/B of library variants on the POOL scenes in one gpurun call: bash tools/pool_variants.sh <tag> <variant> [<variant> ...]   (main = libzyg_amd.so)
. "$(dirname "$0")/common.sh"
set +e
TAG="${1:?tag}"; shift
O="$R/gpurun_out/$TAG"; mkdir -p "$O"
for v in "$@"; do
  if [ "$v" = main ]; then unset ZYG_AMD_LIB; else export ZYG_AMD_LIB=$R/variants/lib_$v.so; fi
  for cfg in "glass_precise:--scene cornell_glass --steps 4 --precise 1" "glass_relaxed:--scene cornell_glass --steps 4" \
             "outdoor4k_precise:--scene outdoor --width 3840 --height 2160 --steps 2 --precise 1" "outdoor4k_relaxed:--scene outdoor --width 3840 --height 2160 --steps 2"; do
    name=${cfg%%:*}; args=${cfg#*:}
    timeout 600 python3 bench.py --no-cpu-baseline --no-extra $args > "$O/${v}_$name.json" 2> "$O/${v}_$name.err"
    python3 -c "
import json
d=json.load(open('$O/${v}_$name.json')); k=d['roofline']['kernel_ms']; print('%-10s %-18s %8.1f Msamples/s %s intersect %.1f shade %.1f scatter %.1f' % ('$v', '$name', d['value'], d['config']['arithmetic'], k['wf_intersect'], k['wf_shade'], k['wf_scatter']))" || tail -3 "$O/${v}_$name.err"
  done
done
