#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE in separate runs) into per-kernel HBM-side bytes per launch.

    python3 tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json> [--factor F | --calibration cal.json] [--commit SHA] [--bench-args "..."]

FETCH_SIZE / WRITE_SIZE are reported in KiB (rocprofv3 derived metrics over TCC_EA0_RDREQ / WRREQ).  Per
/opt/skills/guides/MI355X_MICROARCH.md (HBM section) gfx950's FETCH_SIZE tallies the 128 B requests of wide streaming loads at
64 B, and "other access widths are uncalibrated".  These kernels gather 16 B records out of 256 B blocks, so the factor comes
from tools/calibrate_fetch.py's measurement on that shape (profiles/r02_fetch_calibration.json, stride 256); --factor overrides.
WRITE_SIZE is exact for 16 B/lane stores.  The output names the source commit, the hash of the sources the library was built from
(zyg_amd.build.source_hash() and device_hash() of the tree the passes ran in) and the bench configuration it was taken with: bench.py reports it
as `roofline.traffic` only for that configuration and those sources."""
import argparse
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zyg_amd import build  # noqa: E402


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


ap = argparse.ArgumentParser()
ap.add_argument("fetch_dir")
ap.add_argument("write_dir")
ap.add_argument("out")
ap.add_argument("--factor", type=float, default=None)
ap.add_argument("--calibration", default=None)
ap.add_argument("--commit", default=None)
ap.add_argument("--config", default=None, help="JSON object: the bench configuration (bench.py committed_traffic)")
a = ap.parse_args()
factor = a.factor
if factor is None and a.calibration:
    factor = json.load(open(a.calibration))["strides"]["256"]["fetch_size_factor"]
if factor is None:
    factor = 2.0
fetch, write = load(a.fetch_dir, 'FETCH_SIZE'), load(a.write_dir, 'WRITE_SIZE')
out = {"_meta": {"fetch_size_factor": factor, "source_commit": a.commit, "source_hash": build.source_hash(), "device_hash": build.device_hash(), "config": json.loads(a.config) if a.config else None}}
for k in sorted(set(fetch) | set(write)):
    if not ('wf_' in k or 'film' in k):
        continue
    fb, fl = fetch.get(k, [0.0, 0])
    wb, wl = write.get(k, [0.0, 0])
    launches = max(fl, wl, 1)
    out[k] = {"launches": launches, "fetch_bytes_per_launch_raw": fb / launches, "fetch_bytes_per_launch": factor * fb / launches,
              "write_bytes_per_launch": wb / launches, "hbm_bytes_per_launch": (factor * fb + wb) / launches}
json.dump(out, open(a.out, 'w'), indent=1)
for k, v in out.items():
    if k != "_meta":
        print(f"{k:42s} launches {v['launches']:5d}  fetch x{factor:.2f} {v['fetch_bytes_per_launch'] / 1e6:10.2f} MB  write {v['write_bytes_per_launch'] / 1e6:10.2f} MB")
