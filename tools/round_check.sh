# usage: bash tools/round_check.sh <tag> <commit>: the -m gpu suite, the default bench line, then the profiles of configs[1] and [2]
. "$(dirname "$0")/common.sh"
set +e
TAG="${1:?tag}"
COMMIT="${2:-unknown}"
O="$R/gpurun_out/$TAG"
rm -rf "$O"; mkdir -p "$O"
bash tools/gpu_tests.sh "$TAG/tests"
timeout 900 python3 bench.py > "$O/bench_default.json" 2> "$O/bench_default.err"
cut -c1-1500 "$O/bench_default.json"; tail -3 "$O/bench_default.err"
bash tools/profile_round.sh "${TAG}_cfg1" "$COMMIT" cornell > "$O/profile_cfg1.log" 2>&1
bash tools/profile_round.sh "${TAG}_cfg2" "$COMMIT" cornell_mesh > "$O/profile_cfg2.log" 2>&1
tail -25 "$O/profile_cfg1.log" | cut -c1-300; tail -25 "$O/profile_cfg2.log" | cut -c1-300
