#!/usr/bin/env python3
"""Condenses the rocprofv3 PMC passes of scripts/profile_round.sh into one JSON (and a readable text summary on stdout):
per kernel (full name, grid, workgroup): mean, min, max and dispatch count of every collected counter.  rocprofv3 sums a counter
over its hardware instances, so SQ_* values are chip totals and GRBM_GUI_ACTIVE is the sum over the 8 XCDs.
usage: make_pmc_json.py <prof dir> <out json>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

prof, outp = sys.argv[1], sys.argv[2]
KEEP = ("k_diagt", "k_transitions", "k_hiert", "k_dense", "k_logit", "k_init", "k_pool", "k_pooled", "k_bm_close", "k_chain")
agg = defaultdict(lambda: defaultdict(list))          # (kernel, grid, wg) -> counter -> values
meta = {}
for f in sorted(glob.glob(os.path.join(prof, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            name = r["Kernel_Name"]
            if not any(k in name for k in KEEP):
                continue
            key = (name, int(r["Grid_Size"]), int(r["Workgroup_Size"]))
            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[key] = {"vgpr": int(r["VGPR_Count"]), "accum_vgpr": int(r["Accum_VGPR_Count"]), "sgpr": int(r["SGPR_Count"]),
                         "lds": int(r["LDS_Block_Size"]), "scratch": int(r["Scratch_Size"])}
rows = []
for key, cs in sorted(agg.items(), key=lambda kv: -len(next(iter(kv[1].values())))):
    name, grid, wg = key
    # Jobs whose kernel family is decided on the device enqueue BOTH families per launch and the one the decision does not name returns at
    # once (klara_diagt.h KAuto): such an idle dispatch carries the working kernel's name and grid but ~1e-4 of its counts, and averaged in it
    # pulls every mean down.  A counter's values below 5 % of its maximum are those early returns: dropped, and counted.
    counters = {}
    for c, v in sorted(cs.items()):
        mx = max(v)
        kept = [x for x in v if x >= 0.05 * mx] if mx > 0 else list(v)
        ks = sorted(kept)
        counters[c] = {"mean": sum(kept) / len(kept), "median": ks[len(ks) // 2], "min": ks[0], "max": ks[-1], "n": len(kept),
                       "n_dropped_early_returns": len(v) - len(kept)}
    rows.append({"kernel": name, "grid": grid, "workgroup": wg, **meta[key], "counters": counters})
bench_cfg = {}
try:
    line = json.loads(open(os.path.join(prof, "bench_line_sq.json")).read())
    bench_cfg = {k: line["config"][k] for k in ("steps_per_launch", "nchains_per_gpu", "ndims", "save_rule")}
except Exception as exc:
    bench_cfg = {"error": repr(exc)}
# durations of the headline kernel in its one-shape trace pass (pass 0a): what bench.py's roofline.launch_us has to agree with
trace = {}
for f in glob.glob(os.path.join(prof, "trace_headline", "**", "*kernel_trace.csv"), recursive=True):
    durs = defaultdict(list)
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if "k_diagt<" in r["Kernel_Name"]:
                durs[(r["Kernel_Name"], int(r.get("Grid_Size") or r["Grid_Size_X"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for (name, grid), v in durs.items():
        v.sort()
        trace[f"{name} grid={grid}"] = {"n": len(v), "mean_ns": sum(v) / len(v), "median_ns": v[len(v) // 2], "min_ns": v[0], "max_ns": v[-1]}
try:
    trace["bench_line_launch_us_in_the_profiled_run"] = json.loads(open(os.path.join(prof, "bench_line_trace_headline.json")).read())["roofline"]["launch_us"]
except Exception:
    pass
# registers / scratch of the kernels as loaded in the counter run (bench.py asks the library: klara_get_kernel_attributes) — bench.py compares
# them with what is loaded when it prices a kernel with these counters
loaded = {}
try:
    line = json.loads(open(os.path.join(prof, "bench_line_sq.json")).read())
    for rf in [line.get("roofline", {})] + [v for v in line.get("extra", {}).values() if isinstance(v, dict)]:
        pm = rf.get("pmc") or {}
        if pm.get("key") and pm.get("loaded_vgpr") is not None:      # (every roofline object of the bench line carries one)
            loaded[pm["key"]] = [pm["loaded_vgpr"], pm["loaded_scratch"]]
except Exception as exc:
    loaded = {"error": repr(exc)}
out = {"bench_config": bench_cfg, "loaded_kernel_attributes": loaded, "headline_kernel_trace": trace, "source": "rocprofv3 --pmc passes of scripts/profile_round.sh (bench.py --steps 64 --warmup 16 --reps 1 --no-cpu-baseline), one counter "
                 "group per run, --kernel-trace only; means per WORKING dispatch (a counter's values below 5 % of its maximum — the early returns of the idle sibling of device-decided launches — are dropped and counted in n_dropped_early_returns), chip totals (GRBM_GUI_ACTIVE: sum over the 8 XCDs); "
                 "FETCH_SIZE / WRITE_SIZE in KiB as reported (FETCH_SIZE is doubled by the reader, see profiles/README.md)",
       "kernels": rows}
json.dump(out, open(outp, "w"), indent=1)
for r in rows:
    print(f"{r['kernel'][:110]}  grid={r['grid']} wg={r['workgroup']} vgpr={r['vgpr']} scratch={r['scratch']} lds={r['lds']}")
    for c, v in r["counters"].items():
        print(f"    {c:28s} mean {v['mean']:18.1f}   (n = {v['n']})")
