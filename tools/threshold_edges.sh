# usage: bash tools/threshold_edges.sh <tag>: configs[1] with the two boxes at 12 / 18 / 108 / 972 triangles each and with 9 / 17 props:
# what a scene just past each of the kernels' thresholds costs (small_tris: meshes of <= 16 triangles; the prop tables: <= 16
# props; the wide BLAS: >= 4096 triangles)
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
d=json.load(open(sys.argv[1])); k=d['roofline']['kernel_ms']; print('%-28s %8.1f Msamples/s  trace %.1f shade %.1f scatter %.1f' % (sys.argv[2], d['value'], k['wf_intersect'], k['wf_shade'], k['wf_scatter']))" "$O/$name.json" "$name" || tail -3 "$O/$name.err"
}
run boxes_12
run boxes_18 --box-triangles 18
run boxes_108 --box-triangles 108
run boxes_972 --box-triangles 972
run props_17 --extra-props 8
run props_15 --extra-props 6
run boxes_12_again
