#!/usr/bin/env python3
"""Per-kernel sums of one rocprofv3 --pmc pass (counter_collection.csv), printed one line per kernel.

    python3 tools/pmc_summary.py <pass dir> [--json out.json --commit SHA --config '{...}']

--json writes the sums with the bench configuration, the source commit and zyg_amd.build.source_hash() and device_hash() of the tree the pass
ran in (bench.py quotes profiles/pmc_sq.json only for the same sources and arguments)."""
import argparse
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument("dir")
ap.add_argument("--json", default=None)
ap.add_argument("--commit", default=None)
ap.add_argument("--config", default=None)
a = ap.parse_args()
f = glob.glob(a.dir + '/*/*counter_collection.csv') + glob.glob(a.dir + '/*counter_collection.csv')
rows = list(csv.DictReader(open(f[0])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    k = r['Kernel_Name']
    k = k[k.find('wf_'):] if 'wf_' in k else k
    k = k.split('(')[0]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    agg[k]['_vgpr'] = float(r['VGPR_Count'])
    agg[k]['_scratch'] = float(r['Scratch_Size'])
for k, d in sorted(agg.items()):
    if 'wf_' in k or 'film' in k:
        print(k[:28], {n: f'{v:.3g}' for n, v in sorted(d.items())})
if a.json:
    from zyg_amd import build
    out = {"_meta": {"source_commit": a.commit, "source_hash": build.source_hash(), "device_hash": build.device_hash(), "config": json.loads(a.config) if a.config else None}}
    for k, d in sorted(agg.items()):
        if 'wf_' in k or 'film' in k:
            out[k] = dict(d)
    json.dump(out, open(a.json, "w"), indent=1)
