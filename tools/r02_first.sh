cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02a
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
timeout 300 python3 bench.py --steps 8 --warmup 1 --precise 1 --cpu-seconds 6 > $O/bench_precise.json 2> $O/bench_precise.err
timeout 300 python3 bench.py --steps 8 --warmup 1 --precise 0 --cpu-seconds 6 > $O/bench_relaxed.json 2> $O/bench_relaxed.err
timeout 300 python3 bench.py --steps 4 --warmup 1 --precise 1 --no-cpu-baseline --scene cornell_mesh > $O/bench_mesh_precise.json 2> $O/bench_mesh_precise.err
timeout 300 python3 bench.py --steps 4 --warmup 1 --precise 0 --no-cpu-baseline --scene cornell_mesh > $O/bench_mesh_relaxed.json 2> $O/bench_mesh_relaxed.err
tail -3 $O/pytest.log; cat $O/bench_*.json | cut -c1-400
