#!/bin/bash
# round 4 closing pass on the 64-bit normals: full GPU suite + smoke, the round profile (rocprofv3 stats + PMC), then the two bench lines priced with
# THAT profile's counters (the summary is copied over profiles/r4_pmc_kernels.json on the box before bench.py runs)
cd "$(dirname "$0")/.."
out=gpurun_out/r4_final4; mkdir -p $out
if [ -z "$SKIP_TESTS" ]; then
python -m pytest tests -m gpu -q > $out/pytest.log 2>&1
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $out/pytest.log | tail -6
fi
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -2
bash scripts/profile_round.sh r4c > $out/profile.log 2>&1
[ -s gpurun_out/prof_r4c/pmc_kernels.json ] && cp gpurun_out/prof_r4c/pmc_kernels.json profiles/r4_pmc_kernels.json
python bench.py --steps 20 --warmup 5 > $out/bench_driver_flags.json 2> $out/bench_driver_flags.err
python bench.py > $out/bench_default.json 2> $out/bench_default.err
python - <<'P'
import json
for f in ("bench_driver_flags","bench_default"):
    d=json.loads([l for l in open(f"gpurun_out/r4_final4/{f}.json") if l.startswith("{")][0])
    print(f, "%.4g"%d["value"], d["ms_per_step"], "frac", round(d["roofline"]["frac"],4), "util", d["roofline"].get("utilisation"), "ion", d["roofline"].get("issued_over_necessary"), "cpu", d.get("cpu_baseline",{}).get("value"))
    for k,v in d.get("extra",{}).items():
        if isinstance(v,dict): print("   ",k,{kk:(round(vv,4) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in ("frac","utilisation","issued_over_necessary","frac_lockstep","mfma_pipe_busy_frac")}, "stale" if v.get("pmc",{}).get("stale") else "")
        elif isinstance(v,float): print("   ",k,"%.4g"%v)
P
