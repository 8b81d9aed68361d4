# A/B of one zygpu option through bench.py in one gpurun call (same box): usage: bash tools/ab_option.sh <bench flag> <value a> <value b> [repeats] [more bench args]
. "$(dirname "$0")/common.sh"
set +e
FLAG="${1:?bench flag, e.g. --sort-alive}"; A="$2"; B="$3"; N="${4:-2}"; shift 4 || shift $#
O="$R/gpurun_out/ab"; mkdir -p "$O"
for r in $(seq 1 "$N"); do
  for v in "$A" "$B"; do
    for cfg in "cornell:" "mesh:--scene cornell_mesh" ; do
      tag=${cfg%%:*}; args=${cfg#*:}
      timeout 300 python3 bench.py --steps 8 --no-cpu-baseline --no-extra $FLAG "$v" $args "$@" > "$O/${v}_$tag.json" 2> "$O/${v}_$tag.err"
      python3 -c "
import json
d=json.load(open('$O/${v}_$tag.json')); k=d['roofline']['kernel_ms']; print('%s=%-4s %-8s %8.1f  trace %.1f shade %.1f scatter %.1f shadow %.1f  rms %s' % ('$FLAG','$v','$tag', d['value'], k['wf_intersect'], k['wf_shade'], k['wf_scatter'], k['wf_shadow'], d.get('rms_vs_oracle')))" || tail -3 "$O/${v}_$tag.err"
    done
  done
done
