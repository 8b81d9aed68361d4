# usage: [ZYG_EXP_SCENE=cornell_mound] bash tools/mesh_experiments.sh <tag> variant1 variant2 ...: configs[2] with the main library:
# mesh kernels on (wf_scatter_h + wf_shadow_u) / off (wf_scatter), with wf_mesh_closest, the staged rounds; then each variant
# library (mesh kernels on); parity of the mesh-kernel path against the oracle
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
d=json.load(open(sys.argv[1])); k=d['roofline']['kernel_ms']; print('%-30s %8.1f Msamples/s  trace %.1f shade %.1f scatter %.1f nee %.1f shadow %.1f extend %.1f' % (sys.argv[2], d['value'], k['wf_intersect'], k['wf_shade'], k['wf_scatter'], k['wf_nee'], k['wf_shadow'], k['wf_extend']), d.get('rms_vs_oracle', {}).get('rms', ''))" "$1" "$2" || tail -3 "${1%.json}.err"
}
unset ZYG_AMD_LIB
run() {  # name, extra args
  name="$1"; shift
  timeout 300 python3 bench.py $B "$@" > "$O/$name.json" 2> "$O/$name.err"; show "$O/$name.json" "$SCENE $name"
}
run mk1
run mk0 --mesh-kernels 0
run mk1_again
run mk1_closest --mesh-closest 1
run staged --nee staged
for v in "$@"; do
  export ZYG_AMD_LIB="$R/variants/lib_$v.so"
  run "${v}_mk1"
done
unset ZYG_AMD_LIB
timeout 600 python3 bench.py --steps 4 --no-extra --precise 0 --scene $SCENE --cpu-seconds 5 > "$O/parity.json" 2> "$O/parity.err"; show "$O/parity.json" "$SCENE parity mk1"
