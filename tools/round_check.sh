# usage: bash tools/round_check.sh <tag> <commit>: the -m gpu suite, the default bench line, then the profiles of configs[1], configs[2],
# configs[4]'s scene at 3840x2160 (with configs[2]'s sphere, and the lighter mound variant), the glass Cornell box, the mound and the
# Cornell box with procedural textures; then, in the container:
# python3 tools/collect_profiles.py <tag> <round prefix> copies the summaries into profiles/
. "$(dirname "$0")/common.sh"
set +e
TAG="${1:?tag}"
COMMIT="${2:-unknown}"
O="$R/gpurun_out/$TAG"
rm -rf "$O"; mkdir -p "$O"
bash tools/gpu_tests.sh "$TAG/tests"
timeout 1200 python3 bench.py > "$O/bench_default.json" 2> "$O/bench_default.err"
cut -c1-1500 "$O/bench_default.json"; tail -3 "$O/bench_default.err"
bash tools/profile_round.sh "${TAG}_cfg1" "$COMMIT" cornell > "$O/profile_cfg1.log" 2>&1
bash tools/profile_round.sh "${TAG}_cfg2" "$COMMIT" cornell_mesh > "$O/profile_cfg2.log" 2>&1
ZYG_BENCH_ARGS="--width 3840 --height 2160" ZYG_PROF_STEPS=2 bash tools/profile_round.sh "${TAG}_cfg4" "$COMMIT" outdoor > "$O/profile_cfg4.log" 2>&1
ZYG_BENCH_ARGS="--width 3840 --height 2160" ZYG_PROF_STEPS=2 bash tools/profile_round.sh "${TAG}_cfg4m" "$COMMIT" outdoor_mound > "$O/profile_cfg4m.log" 2>&1
bash tools/profile_round.sh "${TAG}_glass" "$COMMIT" cornell_glass > "$O/profile_glass.log" 2>&1
bash tools/profile_round.sh "${TAG}_mound" "$COMMIT" cornell_mound > "$O/profile_mound.log" 2>&1
ZYG_BENCH_ARGS="--material procedural" bash tools/profile_round.sh "${TAG}_proc" "$COMMIT" cornell > "$O/profile_proc.log" 2>&1
for c in cfg1 cfg2 cfg4 cfg4m glass mound proc; do echo "== $c"; tail -22 "$O/profile_$c.log" | cut -c1-260; done
