#!/usr/bin/env python3
"""FETCH_SIZE calibration on the library's own access shape (16 B records at a fixed stride, one record per lane).

    python3 tools/calibrate_fetch.py                      # timings only -> JSON on stdout
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d DIR -o fetch -- python3 tools/calibrate_fetch.py
    python3 tools/calibrate_fetch.py --summarise DIR OUT.json [timings.json]

The question: does a 16 B read out of a 256 B block cost HBM one 64 B sector or one 128 B line, and
does FETCH_SIZE count it as 64 B either way?  Timing answers the first half: with the buffer far beyond the Infinity Cache a
launch is bandwidth-bound, so ms per record at stride 128 vs stride 64 is 2 if whole 128 B lines move and 1 if 64 B sectors
do.  FETCH_SIZE per record answers the second.  factor = (bytes that really move) / (FETCH_SIZE bytes), per stride."""
import collections
import csv
import ctypes
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
STRIDES = (16, 64, 128, 256, 512)
RECORDS = 8 << 20  # 8 Mi records: 128 MiB .. 4 GiB of buffer


def run():
    import zyg_amd
    lib = zyg_amd.load_library()
    ctx = ctypes.c_void_p()
    assert lib.zygpu_create(0, ctypes.byref(ctx)) == 0, lib.zygpu_last_error()
    lib.zygpu_calibrate_gather.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(ctypes.c_double)]
    out = {}
    for s in STRIDES:
        ms = ctypes.c_double()
        assert lib.zygpu_calibrate_gather(ctx, s, RECORDS, 8, ctypes.byref(ms)) == 0, lib.zygpu_last_error()
        out[str(s)] = {"ms_per_launch": ms.value, "ns_per_record": ms.value * 1e6 / RECORDS, "records": RECORDS,
                       "requested_gbs": RECORDS * 16 / (ms.value * 1e-3) / 1e9, "span_gbs": RECORDS * s / (ms.value * 1e-3) / 1e9}
    lib.zygpu_destroy(ctx)
    print(json.dumps(out))


def summarise(d, out_path, timings_path=None):
    f = glob.glob(d + "/*counter_collection.csv") + glob.glob(d + "/*/*counter_collection.csv")
    per = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == "FETCH_SIZE" and "calib_gather" in r["Kernel_Name"]:
            stride = r["Kernel_Name"].split("<")[1].split(">")[0].rstrip("u")
            per[stride][0] += float(r["Counter_Value"]) * 1024.0
            per[stride][1] += 1
    timings = json.load(open(timings_path)) if timings_path else {}
    out = {"records_per_launch": RECORDS, "strides": {}}
    for stride, (b, n) in sorted(per.items(), key=lambda kv: int(kv[0])):
        e = {"launches": n, "fetch_size_bytes_per_record": b / n / RECORDS}
        e.update(timings.get(stride, {}))
        out["strides"][stride] = e
    t = {k: v.get("ns_per_record") for k, v in out["strides"].items()}
    if t.get("64") and t.get("128") and t.get("256"):
        # bytes per record that move, from the bandwidth-bound timings relative to the dense stride-64 case (64 B per record)
        for k in ("128", "256", "512"):
            if t.get(k):
                moved = 64.0 * t[k] / t["64"]
                out["strides"][k]["moved_bytes_per_record_from_timing"] = moved
                out["strides"][k]["fetch_size_factor"] = moved / out["strides"][k]["fetch_size_bytes_per_record"]
        out["strides"]["64"]["fetch_size_factor"] = 64.0 / out["strides"]["64"]["fetch_size_bytes_per_record"]
    json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--summarise":
        summarise(*sys.argv[2:])
    else:
        run()
