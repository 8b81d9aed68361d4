# kernel-trace stats of one bench run: bash tools/kstats.sh <tag> <bench args...>
. "$(dirname "$0")/common.sh"
set +e
TAG="${1:?tag}"; shift
O="$R/gpurun_out/ks_$TAG"; rm -rf "$O"; mkdir -p "$O"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O" -o kt -- python3 "$R/bench.py" --warmup 0 --no-cpu-baseline --no-extra "$@" > "$O/bench.json" 2> "$O/err.txt"
cd "$R"
cp "$(find "$O" -name "*kernel_stats.csv" | head -1)" "$O/kernel_stats.csv" 2>/dev/null
find "$O" -name "*.csv" -size +2M -delete; find "$O" -name "*.db" -delete
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/kernel_stats.csv")))
for r in rows[:12]:
    n=r["Name"]; n=n[n.find("wf_"):] if "wf_" in n else n
    print("%-70s calls %5s total %8.1f ms avg %8.1f us  %5s %%" % (n[:70], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, r["Percentage"]))
PY
