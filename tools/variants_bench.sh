# usage: bash tools/variants_bench.sh name1 name2 ...   ("main" = zyg_amd/libzyg_amd.so)
. "$(dirname "$0")/common.sh"
set +e
O=$R/gpurun_out/variants
mkdir -p $O
for v in "$@"; do
  if [ "$v" = main ]; then unset ZYG_AMD_LIB; else export ZYG_AMD_LIB=$R/variants/lib_$v.so; fi
  for cfg in "relaxed:--precise 0" "precise:--precise 1" "mesh:--precise 0 --scene cornell_mesh"; do
    tag=${cfg%%:*}; args=${cfg#*:}
    timeout 300 python3 bench.py --steps 8 --no-cpu-baseline --no-extra $args ${ZYG_BENCH_ARGS:-} > $O/${v}_$tag.json 2> $O/${v}_$tag.err
    python3 -c "
import json
d=json.load(open('$O/${v}_$tag.json')); k=d['roofline']['kernel_ms']; print('%-10s %-8s %8.1f  trace %.1f shade %.1f scatter %.1f' % ('$v','$tag', d['value'], k['wf_intersect'], k['wf_shade'], k['wf_scatter']))" || tail -3 $O/${v}_$tag.err
  done
done
