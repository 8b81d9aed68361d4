# rocprofv3 PMC passes for the latency / instruction-mix questions (one pass per counter group; no trace domains with --pmc)
# usage: bash tools/pmc_latency.sh <tag> [scene] [passes file: one counter group per line]   -> gpurun_out/lat_<tag>/{avail.txt,pass*.txt}
. "$(dirname "$0")/common.sh"
set +e
TAG="${1:?tag}"
SCENE="${2:-cornell}"
O="$R/gpurun_out/lat_$TAG"
rm -rf "$O"; mkdir -p "$O"
B="--no-cpu-baseline --no-extra --scene $SCENE --steps 2 --warmup 0 ${ZYG_BENCH_ARGS:-}"
cd /tmp
rocprofv3 --list-avail > "$O/avail_all.txt" 2>&1
grep -o "SQ_[A-Z_0-9]*\|TCP_[A-Z_0-9a-z]*\|TA_[A-Z_0-9a-z]*" "$O/avail_all.txt" | sort -u > "$O/avail.txt"
n=0
while read -r counters; do
  n=$((n + 1))
  rocprofv3 --output-format csv --pmc $counters -d "$O/p$n" -o p -- python3 "$R/bench.py" $B > /dev/null 2> "$O/p$n.err"
  python3 "$R/tools/pmc_summary.py" "$O/p$n" > "$O/pass$n.txt" 2>&1
  find "$O/p$n" -name "*.csv" -size +2M -delete; find "$O/p$n" -name "*.db" -delete
  echo "== $counters"; grep "wf_scatter<false, false" "$O/pass$n.txt" | cut -c1-600
  grep "wf_trace_shade<false, false, false" "$O/pass$n.txt" | cut -c1-600
done < "${3:-$R/tools/pmc_passes_sq.txt}"
