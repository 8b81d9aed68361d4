# usage: bash tools/mesh_experiments.sh <tag> variant1 variant2 ...: configs[2] (the displaced sphere) with the main library in-place
# and staged NEE, then each variant library in-place; the section profile when variants/lib_prof.so exists
. "$(dirname "$0")/common.sh"
set +e
TAG="${1:?tag}"; shift
O="$R/gpurun_out/$TAG"
mkdir -p "$O"
B="--steps 4 --no-cpu-baseline --no-extra --precise 0 --scene cornell_mesh"
show() {
  python3 -c "
import json,sys
d=json.load(open(sys.argv[1])); k=d['roofline']['kernel_ms']; print('%-22s %8.1f Msamples/s  trace %.1f shade %.1f scatter %.1f nee %.1f shadow %.1f extend %.1f' % (sys.argv[2], d['value'], k['wf_intersect'], k['wf_shade'], k['wf_scatter'], k['wf_nee'], k['wf_shadow'], k['wf_extend']))" "$1" "$2" || tail -3 "${1%.json}.err"
}
unset ZYG_AMD_LIB
for rep in 1 2; do
  timeout 300 python3 bench.py $B > "$O/main_$rep.json" 2> "$O/main_$rep.err"; show "$O/main_$rep.json" "main in-place #$rep"
done
timeout 300 python3 bench.py $B --nee staged > "$O/main_staged.json" 2> "$O/main_staged.err"; show "$O/main_staged.json" "main staged"
timeout 300 python3 bench.py $B --fuse 0 > "$O/main_unfused.json" 2> "$O/main_unfused.err"; show "$O/main_unfused.json" "main unfused"
for v in "$@"; do
  export ZYG_AMD_LIB="$R/variants/lib_$v.so"
  timeout 300 python3 bench.py $B > "$O/$v.json" 2> "$O/$v.err"; show "$O/$v.json" "$v"
done
if [ -f "$R/variants/lib_prof.so" ]; then
  export ZYG_AMD_LIB="$R/variants/lib_prof.so"
  ZYG_SCENE=mesh timeout 300 python3 tools/sections.py > "$O/sections_mesh.txt" 2>&1; cat "$O/sections_mesh.txt"
  timeout 300 python3 tools/sections.py > "$O/sections_cornell.txt" 2>&1; cat "$O/sections_cornell.txt"
fi
unset ZYG_AMD_LIB
for s in glass outdoor; do timeout 600 python3 tools/relaxed_diag.py $s > "$O/relaxed_diag_$s.txt" 2>&1; cat "$O/relaxed_diag_$s.txt"; done
timeout 600 python3 tools/ray_diag.py > "$O/ray_diag.txt" 2>&1; cat "$O/ray_diag.txt"
