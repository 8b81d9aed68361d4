cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/r02j
rm -rf $O; mkdir -p $O
timeout 1200 python3 -m pytest tests -m gpu -q -k "$1" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
grep -E "^FAILED|passed|failed|^E  " $O/pytest.log | head -60
