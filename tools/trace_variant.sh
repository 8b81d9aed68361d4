# kernel trace of one bench step with a variant library: bash tools/trace_variant.sh <variant|main> [bench args]  -> gpurun_out/kt_<variant>/
. "$(dirname "$0")/common.sh"
set +e
V="${1:?variant}"; shift
if [ "$V" != main ]; then export ZYG_AMD_LIB=$R/variants/lib_$V.so; fi
O="$R/gpurun_out/kt_$V"; rm -rf "$O"; mkdir -p "$O"
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d "$O" -o kt -- python3 "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-extra "$@" > "$O/bench.json" 2> "$O/err.txt"
cd "$R"
for k in "wf_scatter<false, false" "wf_trace_shade<false, false, false" "wf_trace_shade<false, false, true" "wf_rscatter" "wf_rtrace" "wf_rgenerate"; do echo "$V $k:"; python3 tools/launch_times.py "$O" "$k" 20; done
find "$O" -name "*.csv" -size +2M -delete
