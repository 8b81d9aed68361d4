# A/B of the HSA runtime's scratch knobs (environment of the process) on configs[1], the outdoor scene and the glass box, one gpurun call
. "$(dirname "$0")/common.sh"
set +e
run() {
  a=$(env $1 python3 bench.py --no-extra --no-cpu-baseline --steps 8 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])")
  b=$(env $1 python3 bench.py --scene outdoor_mound --width 3840 --height 2160 --no-extra --no-cpu-baseline --steps 2 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])")
  c=$(env $1 python3 bench.py --scene cornell_glass --no-extra --no-cpu-baseline --steps 6 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])")
  d=$(env $1 python3 bench.py --scene cornell_mesh --no-extra --no-cpu-baseline --steps 6 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])")
  echo "$1: configs1 $a outdoor_mound $b glass $c configs2 $d"
}
run ZYG_NONE=1
run HSA_NO_SCRATCH_THREAD_LIMITER=1
run HSA_SCRATCH_SINGLE_LIMIT=4294967296
run HSA_SCRATCH_SINGLE_LIMIT_ASYNC=17179869184
run HSA_NO_SCRATCH_RECLAIM=1
run HSA_ENABLE_SCRATCH_ASYNC_RECLAIM=0
run ZYG_NONE=2
