# usage: bash tools/variants_mesh.sh name1 name2 ...   ("main" = zyg_amd/libzyg_amd.so): configs[2] (the 1 M-triangle mesh) alone
. "$(dirname "$0")/common.sh"
set +e
O=$R/gpurun_out/variants_mesh
mkdir -p $O
for v in "$@"; do
  if [ "$v" = main ]; then unset ZYG_AMD_LIB; else export ZYG_AMD_LIB=$R/variants/lib_$v.so; fi
  timeout 300 python3 bench.py --steps 8 --no-cpu-baseline --no-extra --precise 0 --scene cornell_mesh > $O/${v}.json 2> $O/${v}.err
  python3 -c "
import json
d=json.load(open('$O/${v}.json')); k=d['roofline']['kernel_ms']; print('%-10s %8.1f  trace %.1f shade %.1f scatter %.1f  rms %s' % ('$v', d['value'], k['wf_intersect'], k['wf_shade'], k['wf_scatter'], d.get('rms_vs_oracle', d.get('parity'))))" || tail -3 $O/${v}.err
done
