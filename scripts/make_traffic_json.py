#!/usr/bin/env python3
"""Builds profiles/<tag>_bench_mala_traffic.json from the PMC passes written by scripts/profile_bench.sh.

HBM bytes per launch = FETCH_SIZE x 2 + WRITE_SIZE (KiB), the correction being calibrated in the same run on
k_init (reads X once, writes G and LT) — see the "calibration" field.   usage: make_traffic_json.py <prof dir> <out json>
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

prof, outp = sys.argv[1], sys.argv[2]
NCH, D = 65536, 100


def mean_counter(sub, counter):
    agg = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(prof, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                a = agg[r["Kernel_Name"]]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
    return {k: v[1] / v[0] for k, v in agg.items()}


fetch, write = mean_counter("pmc_fetch", "FETCH_SIZE"), mean_counter("pmc_write", "WRITE_SIZE")
kt = max((k for k in fetch if "k_transitions" in k or "k_diagt<" in k), key=lambda k: fetch[k])
ki = next(k for k in fetch if "k_init<" in k or "k_diagt_init<" in k)
x_kb = NCH * D * 8 / 1024
corr = 2.0
out = {
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes (scripts/profile_bench.sh), bench.py --steps 200 --warmup 20 --streams 1",
    "kernel": kt, "nchains": NCH, "ndims": D, "steps_per_launch": 1,
    "FETCH_SIZE_raw_kb_per_launch": fetch[kt], "WRITE_SIZE_raw_kb_per_launch": write[kt], "fetch_correction": corr,
    "calibration": (f"{ki} (the init kernel) reads X = {x_kb:.0f} KiB and reports FETCH_SIZE {fetch[ki]:.1f} (x{corr} = {fetch[ki] * corr:.1f}); it writes "
                    f"G + LT = {x_kb + NCH * 8 / 1024:.0f} KiB and reports WRITE_SIZE {write[ki]:.1f}. So FETCH_SIZE is doubled "
                    "(MI355X_MICROARCH.md HBM section) and WRITE_SIZE is taken as is, both in KiB."),
    "traffic_bytes_per_launch": (fetch[kt] * corr + write[kt]) * 1024,
}
json.dump(out, open(outp, "w"), indent=1)
print(json.dumps(out, indent=1))
