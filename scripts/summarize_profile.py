#!/usr/bin/env python3
"""Condenses rocprofv3 CSV output (kernel stats + PMC passes) into a short text summary for profiles/."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


print("# rocprofv3 summary:", os.path.basename(out))
for f in find("trace/**/*kernel_stats.csv"):
    print(f"\n## kernel stats ({os.path.relpath(f, out)})")
    with open(f) as fh:
        rows = list(csv.DictReader(fh))
    for r in rows[:12]:
        name = r.get("Name", "")[:90]
        print(f"{name:90s} calls={r.get('Calls')} total_ns={r.get('TotalDurationNs')} avg_ns={r.get('AverageNs')} pct={r.get('Percentage')}")

for tag, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    for f in find(f"{tag}/**/*counter_collection.csv"):
        agg = defaultdict(lambda: [0, 0.0])
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r.get("Counter_Name") != counter:
                    continue
                k = r.get("Kernel_Name", "")[:90]
                agg[k][0] += 1
                agg[k][1] += float(r.get("Counter_Value", 0))
        print(f"\n## {counter} per dispatch ({os.path.relpath(f, out)}) [raw counter units; see MI355X_MICROARCH.md §HBM]")
        for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:10]:
            print(f"{k:90s} dispatches={n} mean={v / max(n, 1):.1f}")
