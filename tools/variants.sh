#!/bin/bash
# runs bench.py once per library variant under variants/ (built by hand with -DZYG_*_WAVES=...) and prints one line each
for so in variants/lib_*.so; do
  ZYG_AMD_LIB=$PWD/$so python3 bench.py --steps 3 --warmup 1 --cpu-seconds 1 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$so', d['value'], d['roofline']['kernel_ms'], d['rms_vs_oracle']['bit_identical_pixels'])"
done
