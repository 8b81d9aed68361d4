# A/B of an environment switch of the library (read at zygpu_create: ZYG_TABLE_COMPLETE, ZYG_FIRST_KEEP) on the outdoor scene at 3840x2160
# and the glass Cornell box, in one gpurun call: bash tools/ab_env.sh ZYG_FIRST_KEEP
. "$(dirname "$0")/common.sh"
set +e
for v in 1 0 1 0; do
  a=$(env $1=$v python3 bench.py --scene outdoor_mound --width 3840 --height 2160 --no-extra --no-cpu-baseline --steps 2 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])")
  b=$(env $1=$v python3 bench.py --scene outdoor --width 3840 --height 2160 --no-extra --no-cpu-baseline --steps 2 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])")
  c=$(env $1=$v python3 bench.py --scene cornell_glass --no-extra --no-cpu-baseline --steps 6 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])")
  echo "$1=$v outdoor_mound $a outdoor $b glass $c"
done
