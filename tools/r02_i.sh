cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02i
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_traversal.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
grep -E "^FAILED|passed|failed|^E  " $O/pytest.log | head -20
for cfg in "relaxed:--precise 0" "relaxed_nofuse:--precise 0 --fuse-tick 0" "precise:--precise 1" "mesh:--precise 0 --scene cornell_mesh" "mesh_nofuse:--precise 0 --scene cornell_mesh --fuse-tick 0" "subst:--precise 0 --material substitute" "subst_nofuse:--precise 0 --material substitute --fuse-tick 0"; do
  tag=${cfg%%:*}; args=${cfg#*:}
  timeout 300 python3 bench.py --steps 8 --no-cpu-baseline --no-extra $args > $O/$tag.json 2> $O/$tag.err
  python3 -c "
import json
d=json.load(open('$O/$tag.json')); k=d['roofline']['kernel_ms']; print('%-16s %8.1f  trace %.1f shade %.1f scatter %.1f' % ('$tag', d['value'], k['wf_intersect'], k['wf_shade'], k['wf_scatter']))" || tail -3 $O/$tag.err
done
