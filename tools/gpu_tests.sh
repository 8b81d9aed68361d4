# usage: bash tools/gpu_tests.sh <out tag> [pytest -k expression]: the -m gpu tests into gpurun_out/<tag>/pytest.log
. "$(dirname "$0")/common.sh"
O="$R/gpurun_out/${1:?out tag}"
rm -rf "$O"; mkdir -p "$O"
K="${2:-}"
if [ -n "$K" ]; then
  timeout 2400 python3 -m pytest tests -m gpu -q -s -k "$K" > "$O/pytest.log" 2>&1 && rc=0 || rc=$?
else
  timeout 2400 python3 -m pytest tests -m gpu -q --durations=25 > "$O/pytest.log" 2>&1 && rc=0 || rc=$?
fi
echo "pytest rc $rc" >> "$O/pytest.log"
grep -E "^FAILED|^ERROR|passed|failed|^E  |relaxed vs oracle|pytest rc" "$O/pytest.log" | head -60
