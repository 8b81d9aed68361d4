# bench.py with the regenerating kernels at several pool sizes, next to the tick loop, in one gpurun call
. "$(dirname "$0")/common.sh"
set +e
O="$R/gpurun_out/regen_sweep"; mkdir -p "$O"
run() {
  tag="$1"; shift
  timeout 300 python3 bench.py --steps 8 --no-cpu-baseline --no-extra "$@" > "$O/$tag.json" 2> "$O/$tag.err"
  python3 -c "
import json
d=json.load(open('$O/$tag.json')); k=d['roofline']['kernel_ms']; pk=d['roofline'].get('per_kernel',{})
print('%-22s %8.1f  trace %.1f shade %.1f scatter %.1f  launches %s  rms %s' % ('$tag', d['value'], k['wf_intersect'], k['wf_shade'], k['wf_scatter'], d['roofline'].get('launches'), (d.get('rms_vs_oracle') or {}).get('rms')))" || tail -3 "$O/$tag.err"
}
run tick "$@"
for ks in ${REGEN_KSLOTS:-1024 2048 4096 8192}; do run regen_$ks --regenerate 1 --regen-kslots $ks "$@"; done
run tick_again "$@"
