cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/evidence
rm -rf $O; mkdir -p $O
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -4 $O/pytest.log
bash tools/profile_round.sh final $1 > $O/profile.log 2>&1
timeout 900 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
cut -c1-600 $O/bench_default.json; tail -2 $O/bench_default.err
