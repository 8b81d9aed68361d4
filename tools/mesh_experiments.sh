# usage: bash tools/mesh_experiments.sh <tag> variant1 variant2 ...: configs[2] (the displaced sphere): the main library with the
# NEE in place / staged and the mesh kernels on / off, then each variant library (staged, mesh kernels on); parity of the staged
# mesh-kernel path against the oracle; the section profile when variants/lib_prof.so exists
. "$(dirname "$0")/common.sh"
set +e
TAG="${1:?tag}"; shift
O="$R/gpurun_out/$TAG"
mkdir -p "$O"
SCENE="${ZYG_EXP_SCENE:-cornell_mesh}"
B="--steps 4 --no-cpu-baseline --no-extra --precise 0 --scene $SCENE"
show() {
  python3 -c "
import json,sys
d=json.load(open(sys.argv[1])); k=d['roofline']['kernel_ms']; print('%-34s %8.1f Msamples/s  trace %.1f shade %.1f scatter %.1f nee %.1f shadow %.1f extend %.1f' % (sys.argv[2], d['value'], k['wf_intersect'], k['wf_shade'], k['wf_scatter'], k['wf_nee'], k['wf_shadow'], k['wf_extend']), d.get('rms_vs_oracle', {}).get('rms', ''))" "$1" "$2" || tail -3 "${1%.json}.err"
}
unset ZYG_AMD_LIB
run() {  # name, extra args
  name="$1"; shift
  timeout 300 python3 bench.py $B "$@" > "$O/$name.json" 2> "$O/$name.err"; show "$O/$name.json" "$name"
}
run main_inplace_mk1
run main_inplace_mk0 --mesh-kernels 0
run main_staged_mk1 --nee staged
run main_staged_mk0 --nee staged --mesh-kernels 0
run main_staged_mk1_again --nee staged
for v in "$@"; do
  export ZYG_AMD_LIB="$R/variants/lib_$v.so"
  run "${v}_staged" --nee staged
  run "${v}_inplace"
done
unset ZYG_AMD_LIB
# parity of the new paths against the oracle (row band)
timeout 600 python3 bench.py --steps 4 --no-extra --precise 0 --scene $SCENE --cpu-seconds 5 --nee staged > "$O/parity_staged.json" 2> "$O/parity_staged.err"; show "$O/parity_staged.json" "parity staged mk1"
timeout 600 python3 bench.py --steps 4 --no-extra --precise 0 --scene $SCENE --cpu-seconds 5 > "$O/parity_inplace.json" 2> "$O/parity_inplace.err"; show "$O/parity_inplace.json" "parity in-place mk1"
if [ -f "$R/variants/lib_prof.so" ] && [ -n "${ZYG_EXP_SECTIONS:-}" ]; then
  export ZYG_AMD_LIB="$R/variants/lib_prof.so"
  ZYG_SCENE=mesh timeout 300 python3 tools/sections.py > "$O/sections_mesh.txt" 2>&1; cat "$O/sections_mesh.txt"
fi
