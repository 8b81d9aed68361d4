for tc in 1 0 1 0; do
  for sc in cornell_glass; do
    ZYG_TABLE_COMPLETE=$tc python bench.py --scene $sc --no-extra --no-cpu-baseline --steps 6 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$sc tc=$tc', d['value'])"
  done
done
for tc in 1 0 1 0; do
  for sc in outdoor outdoor_mound; do
    ZYG_TABLE_COMPLETE=$tc python bench.py --scene $sc --width 3840 --height 2160 --no-extra --no-cpu-baseline --steps 2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$sc tc=$tc', d['value'])"
  done
done
