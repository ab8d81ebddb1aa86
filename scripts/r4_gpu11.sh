#!/bin/bash
cd "$(dirname "$0")/.."
out=gpurun_out/r4_gpu11; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q -k "dense or layout_choice or random_configurations" > $out/pytest.log 2>&1
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $out/pytest.log | tail -8
python bench.py --no-cpu-baseline > $out/bench_default.json 2> $out/bench_default.err
python - <<'P'
import json
d=json.loads([l for l in open("gpurun_out/r4_gpu11/bench_default.json") if l.startswith("{")][0])
print("%.4g"%d["value"], d["ms_per_step"])
for k,v in d.get("extra",{}).items():
    if isinstance(v,dict): print("   ",k,{kk:(round(vv,4) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in ("frac","utilisation","issued_over_necessary","frac_lockstep","achieved")})
    elif isinstance(v,float): print("   ",k,"%.4g"%v)
    else: print("   ",k,v)
P
