cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02b
rm -rf $O; mkdir -p $O
cd $R
timeout 1200 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
timeout 300 python3 tools/cpu_probe.py > $O/cpu_probe.log 2>&1
timeout 600 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 120 python3 tools/calibrate_fetch.py > $O/calib_timings.json 2> $O/calib.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/calib -o fetch -- python3 tools/calibrate_fetch.py > /dev/null 2> $O/calib_prof.err
python3 tools/calibrate_fetch.py --summarise $O/calib $O/fetch_calibration.json $O/calib_timings.json > $O/calib_summary.log 2>&1
find $O -name "*.csv" -size +3M -delete; find $O -name "*.db" -delete
tail -5 $O/pytest.log; cat $O/cpu_probe.log; cut -c1-1500 $O/bench_default.json; tail -3 $O/bench_default.err; cat $O/calib_summary.log | tail -60
