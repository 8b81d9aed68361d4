#!/usr/bin/env python3
"""Prints the durations (us) of the first launches of every wavefront kernel from a rocprofv3 kernel trace CSV."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
seen = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name']
    if 'wf_' not in n or 'true>' in n or 'ELb1E' in n:
        continue
    n = n[n.find('wf_'):].split('(')[0]
    if len(seen[n]) < (int(sys.argv[2]) if len(sys.argv) > 2 else 6):
        seen[n].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000)
for n, v in seen.items():
    print(n, ' '.join(f'{x:.0f}' for x in v))
