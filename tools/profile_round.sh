cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_f
rm -rf $O; mkdir -p $O
cd $R
python3 bench.py --steps 8 --warmup 1 > $O/bench_steps8.json 2> $O/bench_steps8.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 bench.py --steps 4 --warmup 0 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/kt.err
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $O/fetch -o fetch -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline > /dev/null 2> $O/fetch.err
rocprofv3 --output-format csv --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $O/write -o write -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline > /dev/null 2> $O/write.err
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU -d $O/sq -o sq -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline > /dev/null 2> $O/sq.err
python3 tools/pmc_traffic.py $O/fetch $O/write $O/traffic.json > $O/traffic.txt 2>&1
python3 tools/pmc_summary.py $O/sq > $O/pmc_sq.txt 2>&1
python3 tools/pmc_summary.py $O/write > $O/pmc_tcc.txt 2>&1
python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --scene cornell_mesh > $O/bench_cfg2.json 2>/dev/null
python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --material substitute > $O/bench_substitute.json 2>/dev/null
python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --nee staged > $O/bench_staged.json 2>/dev/null
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --scene outdoor > $O/bench_outdoor.json 2>/dev/null
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --scene cornell_glass > $O/bench_glass.json 2>/dev/null
find $O -name "*kernel_stats.csv" | head; ls $O
# keep only small files
find $O -name "*.csv" -size +3M -delete
find $O -name "*.db" -delete
tail -c 600 $O/bench_steps8.json
