#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE in separate runs) into per-kernel HBM-side bytes per launch.

    python3 tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json>

FETCH_SIZE / WRITE_SIZE are reported in KiB (rocprofv3 derived metrics over TCC_EA0_RDREQ / WRREQ).  Corrections per
/opt/skills/guides/MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE tallies the 128 B requests of wide
(16 B/lane) loads at 64 B, so it is doubled; WRITE_SIZE is exact for 16 B/lane stores.  The calibration check the guide
asks for uses wf_generate, whose byte count is known (it only writes: 10 x 16 B + 2 x 8 B + 4 B per path)."""
import collections
import csv
import glob
import json
import sys


def load(d, counter):
    f = glob.glob(d + '/*counter_collection.csv') + glob.glob(d + '/*/*counter_collection.csv')
    per = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f[0])):
        if r['Counter_Name'] != counter:
            continue
        k = r['Kernel_Name']
        k = k[k.find('wf_'):] if 'wf_' in k else k
        k = k.split('(')[0]
        per[k][0] += float(r['Counter_Value']) * 1024.0
        per[k][1] += 1
    return per


fetch, write = load(sys.argv[1], 'FETCH_SIZE'), load(sys.argv[2], 'WRITE_SIZE')
out = {}
for k in sorted(set(fetch) | set(write)):
    if not ('wf_' in k or 'film' in k):
        continue
    fb, fl = fetch.get(k, [0.0, 0])
    wb, wl = write.get(k, [0.0, 0])
    launches = max(fl, wl, 1)
    out[k] = {"launches": launches, "fetch_bytes_per_launch_raw": fb / launches, "fetch_bytes_per_launch": 2.0 * fb / launches,
              "write_bytes_per_launch": wb / launches, "hbm_bytes_per_launch": (2.0 * fb + wb) / launches}
json.dump(out, open(sys.argv[3], 'w'), indent=1)
for k, v in out.items():
    print(f"{k:42s} launches {v['launches']:5d}  fetch x2 {v['fetch_bytes_per_launch'] / 1e6:10.2f} MB  write {v['write_bytes_per_launch'] / 1e6:10.2f} MB")
