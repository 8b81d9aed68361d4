# POOL scenes (Glass): speed in precise / relaxed arithmetic, and relaxed-vs-precise RMS with the boxes on / off the floor.
# usage: bash tools/pool_experiments.sh <tag> [skip_diag]
. "$(dirname "$0")/common.sh"
set +e
TAG="${1:?tag}"; O="$R/gpurun_out/$TAG"; mkdir -p "$O"
run() { # name, bench args
  name=$1; shift
  timeout 600 python3 bench.py --no-cpu-baseline --no-extra "$@" > "$O/$name.json" 2> "$O/$name.err"
  python3 -c "
import json
d=json.load(open('$O/$name.json')); k=d['roofline']['kernel_ms']; print('%-26s %8.1f Msamples/s %s intersect %.1f shade %.1f scatter %.1f' % ('$name', d['value'], d['config']['arithmetic'], k['wf_intersect'], k['wf_shade'], k['wf_scatter']))" || tail -3 "$O/$name.err"
}
run outdoor4k_precise --scene outdoor --width 3840 --height 2160 --steps 2
run outdoor4k_relaxed --scene outdoor --width 3840 --height 2160 --steps 2 --pool-relaxed 1
run outdoor1080_precise --scene outdoor --steps 4
run outdoor1080_relaxed --scene outdoor --steps 4 --pool-relaxed 1
run glass_precise --scene cornell_glass --steps 4
run glass_relaxed --scene cornell_glass --steps 4 --pool-relaxed 1
if [ -z "${2:-}" ]; then
  timeout 900 python3 tools/pool_relaxed_diag.py glass 640 360 64 256 2>&1 | grep -v "^Sky" | tee "$O/diag_glass.txt"
  timeout 900 python3 tools/pool_relaxed_diag.py outdoor 640 360 64 256 2>&1 | grep -v "^Sky" | tee "$O/diag_outdoor.txt"
fi
