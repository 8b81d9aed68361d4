cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02c
rm -rf $O; mkdir -p $O
cd $R
timeout 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "cornell_lambert or full_size or config2 or mesh_heightfield" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
timeout 300 python3 bench.py --steps 8 --no-cpu-baseline --no-extra > $O/bench_relaxed.json 2> $O/bench_relaxed.err
timeout 300 python3 bench.py --steps 8 --no-cpu-baseline --no-extra --precise 1 > $O/bench_precise.json 2> $O/bench_precise.err
timeout 300 python3 bench.py --steps 4 --no-cpu-baseline --no-extra --scene cornell_mesh > $O/bench_mesh.json 2> $O/bench_mesh.err
tail -3 $O/pytest.log; for f in relaxed precise mesh; do python3 -c "
import json,sys
d=json.load(open('$O/bench_$f.json')); print('$f', d['value'], d['roofline']['kernel_ms'])"; done
