#!/bin/bash
# round 3: parity after the merged-leapfrog / target-arithmetic changes, then the model configurations' rates
mkdir -p gpurun_out
( timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 ) > gpurun_out/r3_gputests.log
( timeout 900 python bench.py --steps 256 --warmup 32 --reps 2 --no-cpu-baseline 2>&1 | tail -1 ) > gpurun_out/r3_bench_extras.json
tail -4 gpurun_out/r3_gputests.log
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3_bench_extras.json").read())
print("headline", d["value"], d["ms_per_step"])
for k, v in d["extra"].items():
    if isinstance(v, dict):
        print(k, {a: v[a] for a in ("frac", "launch_us", "mfma_pipe_busy_frac") if a in v})
    else:
        print(k, v)
PY
