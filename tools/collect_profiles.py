#!/usr/bin/env python3
"""Copies what tools/round_check.sh left under gpurun_out/ into profiles/ (tracked): per scene the rocprofv3 kernel-trace summary, the
SQ / TCC counter summaries and the traffic summary, named per round, plus the fixed names bench.py quotes:
profiles/traffic.json + pmc_sq.json (the main scene) and profiles/traffic_<scene>.json + pmc_sq_<scene>.json (the `extra` scenes).

    python3 tools/collect_profiles.py <tag of the round_check call> <round prefix, e.g. r05a>"""
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, prefix = sys.argv[1], sys.argv[2]
out = os.path.join(ROOT, "profiles")
scenes = {"cfg1": None, "cfg2": "cornell_mesh", "cfg4": "outdoor", "cfg4m": "outdoor_mound", "glass": "cornell_glass", "mound": "cornell_mound",
          "proc": "SKIP"}  # (--material procedural: no `extra` entry of bench.py quotes it)
for short, scene in scenes.items():
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}_{short}")
    if not os.path.isdir(src):
        continue
    for name in ("kernel_stats.csv", "pmc_sq.txt", "pmc_tcc.txt", "traffic.json", "bench_under_rocprof.json"):
        p = os.path.join(src, name)
        if os.path.exists(p) and os.path.getsize(p) > 0:
            shutil.copy(p, os.path.join(out, f"{prefix}_{short}_{name}"))
    for name in ("traffic.json", "pmc_sq.json") if "SKIP" != scene else ():
        p = os.path.join(src, name)
        if os.path.exists(p) and os.path.getsize(p) > 0:
            stem, ext = os.path.splitext(name)
            shutil.copy(p, os.path.join(out, name if scene is None else f"{stem}_{scene}{ext}"))
for name in ("bench_default.json",):
    p = os.path.join(ROOT, "gpurun_out", tag, name)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(out, f"{prefix}_{name}"))
print("collected", sorted(f for f in os.listdir(out) if f.startswith(prefix) or f.startswith(("traffic", "pmc_sq"))))
