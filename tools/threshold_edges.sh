# usage: bash tools/threshold_edges.sh <tag>: configs[1] with its two boxes at 12 / 18 / 48 / 108 / 972 triangles each and with
# 16 / 17 / 32 / 33 solid props: what a scene just past each of the kernels' thresholds costs (small_tris: meshes of <= 32
# items; the wide BLAS: the meshes past that; the prop tables: <= 112 props, tools/prop_curve.sh goes past them).  ZYG_WIDE_MIN_TRIANGLES / ZYG_MESH_KERNEL_MIN_TRIANGLES move
# the thresholds for an experiment.
. "$(dirname "$0")/common.sh"
set +e
TAG="${1:?tag}"
O="$R/gpurun_out/$TAG"
mkdir -p "$O"
run() {
  name="$1"; shift
  timeout 300 python3 bench.py --steps 8 --no-cpu-baseline --no-extra --precise 0 "$@" > "$O/$name.json" 2> "$O/$name.err"
  python3 -c "
import json,sys
d=json.load(open(sys.argv[1])); k=d['roofline']['kernel_ms']; print('%-28s %8.1f Msamples/s  trace %.1f shade %.1f scatter %.1f shadow %.1f' % (sys.argv[2], d['value'], k['wf_intersect'], k['wf_shade'], k['wf_scatter'], k['wf_shadow']))" "$O/$name.json" "$name" || tail -3 "$O/$name.err"
}
run boxes_12
run boxes_18 --box-triangles 18
run boxes_48 --box-triangles 48
run boxes_108 --box-triangles 108
run boxes_972 --box-triangles 972
run props_16 --extra-props 9
run props_17 --extra-props 10
run props_32 --extra-props 25
run props_33 --extra-props 26
run boxes_12_again
