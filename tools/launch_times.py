#!/usr/bin/env python3
"""Durations of the successive launches of one kernel from a rocprofv3 --kernel-trace csv: python3 tools/launch_times.py <dir> <kernel substring> [n]"""
import csv, glob, sys
d, sub, n = sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 20
f = (glob.glob(d + "/*/*kernel_trace.csv") + glob.glob(d + "/*kernel_trace.csv"))[0]
rows = [r for r in csv.DictReader(open(f)) if sub in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
print(len(rows), "launches;", " ".join(f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6:.3f}" for r in rows[:n]), "ms")
