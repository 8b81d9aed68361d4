#!/bin/bash
# like variants.sh, for BASELINE configs[2] (1M-triangle mesh)
for so in variants/lib_*.so; do
  ZYG_AMD_LIB=$PWD/$so python3 bench.py --steps 3 --warmup 1 --cpu-seconds 1 --scene cornell_mesh 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$so', d['value'], d['roofline']['kernel_ms'], d['rms_vs_oracle']['bit_identical_pixels'])"
done
