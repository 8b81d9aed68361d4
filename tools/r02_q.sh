cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
bash tools/variants_bench.sh main q4 main q4 2>&1 | grep -E "mesh"
# parity band of the quantised walk against the oracle (bench's own check) 
ZYG_AMD_LIB=$R/variants/lib_q4.so timeout 600 python3 bench.py --steps 2 --scene cornell_mesh --no-cpu-baseline --no-extra > gpurun_out/q4_parity.json 2> gpurun_out/q4_parity.err
python3 -c "
import json; d=json.load(open('gpurun_out/q4_parity.json')); print('q4 parity', d.get('rms_vs_oracle'))"
