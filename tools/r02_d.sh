cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02d
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_units.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -4 $O/pytest.log
bash tools/variants_bench.sh main
