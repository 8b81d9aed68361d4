#!/usr/bin/env python3
"""Fill the free device memory with a bit pattern, release it, then run pytest in this process: a read of pool memory that no kernel of
the render has written meets the pattern instead of a fresh box's zeros.  usage: python3 tools/poison_then_test.py <nan|ones|random> <pytest args>"""
import sys

import pytest
import torch

pattern = sys.argv[1]
free, total = torch.cuda.mem_get_info()
chunks = []
left = int(free * 0.92)
while left > 0:
    n = min(left, 8 << 30)
    t = torch.empty(n // 4, dtype=torch.int32, device="cuda")
    if "nan" == pattern:
        t.fill_(0x7FC00000)
    elif "ones" == pattern:
        t.fill_(-1)
    else:
        t.random_(-(1 << 31), (1 << 31) - 1)
    chunks.append(t)
    left -= n
torch.cuda.synchronize()
print(f"poisoned {sum(c.numel() for c in chunks) * 4 / 2**30:.0f} GiB with {pattern}", flush=True)
del chunks, t
torch.cuda.empty_cache()
torch.cuda.synchronize()
probe = torch.empty((1 << 30) // 4, dtype=torch.int32, device="cuda")  # does a new allocation see the pattern, or does the driver clear it?
print(f"a fresh 1 GiB allocation holds {int((probe != 0).sum())} non-zero words of {probe.numel()}", flush=True)
del probe
torch.cuda.empty_cache()
sys.exit(pytest.main(sys.argv[2:]))
