#!/bin/bash
mkdir -p gpurun_out/r4_gpu15
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r4_gpu15/pytest.log
cat gpurun_out/r4_gpu15/pytest.log
AB_TAG=share python scripts/ab_headline.py 2>&1 | tee gpurun_out/r4_gpu15/ab.txt
python bench.py > gpurun_out/r4_gpu15/bench_default.json 2> gpurun_out/r4_gpu15/bench_default.err; tail -c 400 gpurun_out/r4_gpu15/bench_default.err
python bench.py --steps 20 --warmup 5 > gpurun_out/r4_gpu15/bench_driver.json 2> /dev/null
python - <<'PY'
import json
for f in ["bench_default","bench_driver"]:
    try:
        j=json.loads(open(f"gpurun_out/r4_gpu15/{f}.json").read().strip().splitlines()[-1])
        print(f, j["value"], j["ms_per_step"], j["roofline"]["frac"])
        for k,v in j.get("extra",{}).items():
            if isinstance(v,(int,float)): print("  ",k,v)
    except Exception as e: print(f, "ERR", e)
PY
