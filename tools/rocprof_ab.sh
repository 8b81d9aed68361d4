cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/r02p
rm -rf $O; mkdir -p $O
for v in main main; do
  if [ "$v" = main ]; then unset ZYG_AMD_LIB; else export ZYG_AMD_LIB=$R/variants/lib_$v.so; fi
  rm -rf $O/kt
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 bench.py --steps 4 --warmup 0 --no-cpu-baseline --no-extra > /dev/null 2> $O/kt.err
  echo "== $v"; python3 -c "
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if 'wf_trace_shade<false, false' in n or 'wf_scatter<false, false' in n: print('%-70s calls %4s avg %10.1f us' % (n[10:75], r['Calls'], float(r['AverageNs'])/1e3))
" $(find $O/kt -name "*kernel_stats.csv" | head -1)
done
rm -rf $O/kt
