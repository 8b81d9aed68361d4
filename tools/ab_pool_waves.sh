# A/B of occupancy variants of the POOL kernels (zyg_amd/build.py --variant <name> ... -DZYG_*_WAVES=n) on the outdoor scene at 3840x2160
# and the glass Cornell box, in one gpurun call: bash tools/ab_pool_waves.sh "<variant> ..."
. "$(dirname "$0")/common.sh"
set +e
for v in main $1 main; do
  if [ "$v" = main ]; then unset ZYG_AMD_LIB; else export ZYG_AMD_LIB=$R/variants/lib_$v.so; fi
  a=$(python3 bench.py --scene outdoor_mound --width 3840 --height 2160 --no-extra --no-cpu-baseline --steps 2 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])")
  b=$(python3 bench.py --scene cornell_glass --no-extra --no-cpu-baseline --steps 6 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])")
  echo "$v outdoor_mound $a glass $b"
done
