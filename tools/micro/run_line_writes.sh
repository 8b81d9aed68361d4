# usage (on the GPU box): bash tools/micro/run_line_writes.sh: times the store patterns of tools/micro/line_writes.hip
R="${GRAFT_REPO_ROOT:-/root/repo}"
O="$R/gpurun_out/line_writes_out"; mkdir -p "$O"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 "$R/tools/micro/line_writes.hip" -o /tmp/line_writes 2>/dev/null
for a in "1 100" "97 100" "97 60" "97 30"; do timeout 60 /tmp/line_writes $a; done | tee "$O/times.txt"
