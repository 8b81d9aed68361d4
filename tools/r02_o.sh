cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/r02o
rm -rf $O; mkdir -p $O
for cfg in "outdoor_relaxed:--scene outdoor --precise 0" "outdoor_precise:--scene outdoor --precise 1" "glass_relaxed:--scene cornell_glass --precise 0" "glass_precise:--scene cornell_glass --precise 1"; do
  tag=${cfg%%:*}; args=${cfg#*:}
  timeout 300 python3 bench.py --steps 4 --no-cpu-baseline --no-extra $args > $O/$tag.json 2> $O/$tag.err
  python3 -c "
import json
d=json.load(open('$O/$tag.json')); k=d['roofline']['kernel_ms']; print('%-16s %8.1f  trace %.1f shade %.1f scatter %.1f' % ('$tag', d['value'], k['wf_intersect'], k['wf_shade'], k['wf_scatter']))" || tail -3 $O/$tag.err
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-extra --scene outdoor > /dev/null 2> $O/kt.err
head -12 $(find $O/kt -name "*kernel_stats.csv" | head -1) | cut -c1-170
find $O -name "*.csv" -size +1M -delete; find $O -name "*.db" -delete
