# rocprofv3 evidence: kernel trace + PMC passes of bench.py for one scene, summaries into gpurun_out/prof_<tag>
# usage: [ZYG_BENCH_ARGS="--nee staged"] [ZYG_PROF_STEPS=4] bash tools/profile_round.sh <tag> <source commit> [scene (default cornell)]
# Copy kernel_stats.csv, pmc_sq.txt/json, pmc_tcc.txt and traffic.json from there into profiles/ (named per round); for the default
# scene also as profiles/traffic.json and profiles/pmc_sq.json, which bench.py quotes while the sources stay what they were.
. "$(dirname "$0")/common.sh"
TAG="${1:?tag}"
COMMIT="${2:-unknown}"
SCENE="${3:-cornell}"
O="$R/gpurun_out/prof_$TAG"
rm -rf "$O"; mkdir -p "$O"
B="--no-cpu-baseline --no-extra --scene $SCENE ${ZYG_BENCH_ARGS:-}"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt" -o kt -- python3 "$R/bench.py" --steps ${ZYG_PROF_STEPS:-4} --warmup 0 $B > "$O/bench_under_rocprof.json" 2> "$O/kt.err" || true
rocprofv3 --output-format csv --pmc FETCH_SIZE -d "$O/fetch" -o fetch -- python3 "$R/bench.py" --steps 2 --warmup 0 $B > /dev/null 2> "$O/fetch.err" || true
rocprofv3 --output-format csv --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d "$O/write" -o write -- python3 "$R/bench.py" --steps 2 --warmup 0 $B > /dev/null 2> "$O/write.err" || true
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU -d "$O/sq" -o sq -- python3 "$R/bench.py" --steps 2 --warmup 0 $B > /dev/null 2> "$O/sq.err" || true
cd "$R"
# the key bench.py quotes a profile under: its own bench_config() of these very arguments (every option that changes the launched kernels)
CFG="$(python3 "$R/bench.py" --print-config $B)"
python3 tools/pmc_traffic.py "$O/fetch" "$O/write" "$O/traffic.json" --calibration profiles/r02_fetch_calibration.json --commit "$COMMIT" --config "$CFG" > "$O/traffic.txt" 2>&1 || true
python3 tools/pmc_summary.py "$O/sq" --json "$O/pmc_sq.json" --commit "$COMMIT" --config "$CFG" > "$O/pmc_sq.txt" 2>&1 || true
python3 tools/pmc_summary.py "$O/write" > "$O/pmc_tcc.txt" 2>&1 || true
cp "$(find "$O/kt" -name "*kernel_stats.csv" | head -1)" "$O/kernel_stats.csv" 2>/dev/null || true
find "$O" -name "*.csv" -size +2M -delete
find "$O" -name "*.db" -delete
cat "$O/traffic.txt" || true; head -12 "$O/kernel_stats.csv" || true; cut -c1-400 "$O/pmc_sq.txt" || true
