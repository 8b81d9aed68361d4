cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/r02k
rm -rf $O; mkdir -p $O
ZYG_SCENE=mesh ZYG_AMD_LIB=$R/variants/lib_prof.so timeout 600 python3 tools/sections.py > $O/sections_mesh.txt 2>&1
cat $O/sections_mesh.txt
B="--no-cpu-baseline --no-extra --scene cornell_mesh --steps 2 --warmup 0"
rocprofv3 --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum -d $O/tcc -o tcc -- python3 bench.py $B > /dev/null 2> $O/tcc.err
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD -d $O/sq -o sq -- python3 bench.py $B > /dev/null 2> $O/sq.err
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $O/fetch -o fetch -- python3 bench.py $B > /dev/null 2> $O/fetch.err
python3 tools/pmc_summary.py $O/tcc > $O/pmc_tcc.txt 2>&1
python3 tools/pmc_summary.py $O/sq > $O/pmc_sq.txt 2>&1
python3 tools/pmc_summary.py $O/fetch > $O/pmc_fetch.txt 2>&1
find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete
cut -c1-420 $O/pmc_tcc.txt; cut -c1-420 $O/pmc_sq.txt; cut -c1-300 $O/pmc_fetch.txt
