for sn in 1 0 1 0; do
  a=$(ZYG_AMD_LIB=variants/lib_all.so python3 bench.py --scene outdoor_mound --width 3840 --height 2160 --no-extra --no-cpu-baseline --steps 2 --split-nee $sn 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])")
  c=$(ZYG_AMD_LIB=variants/lib_all.so python3 bench.py --scene cornell_glass --no-extra --no-cpu-baseline --steps 6 --split-nee $sn 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])")
  echo "split_nee=$sn outdoor_mound $a glass $c"
done
