# The POOL scenes' evidence in one gpurun call: per-launch durations of one batch (kernel trace), then tools/profile_round.sh
# (kernel stats + PMC passes) for configs[4]'s scene at 3840x2160 and the glass Cornell box at 1080p.
# usage: bash tools/pool_baseline.sh <tag> <commit>
. "$(dirname "$0")/common.sh"
set +e
TAG="${1:?tag}"; COMMIT="${2:-unknown}"
for cfg in "cfg4:outdoor:--width 3840 --height 2160" "glass:cornell_glass:"; do
  name=${cfg%%:*}; rest=${cfg#*:}; scene=${rest%%:*}; args=${rest#*:}
  O="$R/gpurun_out/kt_${TAG}_$name"; rm -rf "$O"; mkdir -p "$O"
  cd /tmp
  rocprofv3 --kernel-trace --output-format csv -d "$O" -o kt -- python3 "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-extra --scene $scene $args > "$O/bench.json" 2> "$O/err.txt"
  cd "$R"
  for k in "wf_generate" "wf_intersect" "wf_shade<false, true, false" "wf_shade<false, true, true" "wf_shade<false, false, false" "wf_shade<false, false, true" "wf_scatter" "film_accumulate"; do echo "$name $k:"; python3 tools/launch_times.py "$O" "$k" 150; done > "$O/launch_times.txt" 2>&1
  find "$O" -name "*.csv" -size +2M -delete; find "$O" -name "*.db" -delete
  cut -c1-1200 "$O/launch_times.txt"
  ZYG_BENCH_ARGS="$args" ZYG_PROF_STEPS=2 bash tools/profile_round.sh "${TAG}_$name" "$COMMIT" $scene > "$R/gpurun_out/profile_${TAG}_$name.log" 2>&1
  tail -30 "$R/gpurun_out/profile_${TAG}_$name.log" | cut -c1-400
done
