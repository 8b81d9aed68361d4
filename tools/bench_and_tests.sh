cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
bash tools/variants_bench.sh main
O=$R/gpurun_out/r02m
rm -rf $O; mkdir -p $O
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
grep -E "^FAILED|passed|failed|^E  " $O/pytest.log | head -30
