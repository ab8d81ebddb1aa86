#!/bin/bash
mkdir -p gpurun_out/r4_gpu21
for i in 1 2; do
KLARA_HIP_LIB=klara.jl_amd/lib/libklara_hip_rmw.so AB_TAG=fold_rmw python scripts/ab_headline.py 2>&1 | tail -1 | tee -a gpurun_out/r4_gpu21/ab_fold_atomic.txt
AB_TAG=fold_atomic python scripts/ab_headline.py 2>&1 | tail -1 | tee -a gpurun_out/r4_gpu21/ab_fold_atomic.txt
done
for lib in klara.jl_amd/lib/libklara_hip_rmw.so klara.jl_amd/lib/libklara_hip.so; do
KLARA_HIP_LIB=$lib python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$lib driver flags value %.4g ms_per_step %.5f' % (d['value'], d['ms_per_step']), d['config'].get('reps_ms_per_step'))" | tee -a gpurun_out/r4_gpu21/ab_fold_atomic.txt
KLARA_HIP_LIB=$lib python bench.py --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$lib default flags value %.4g ms_per_step %.5f' % (d['value'], d['ms_per_step']))" | tee -a gpurun_out/r4_gpu21/ab_fold_atomic.txt
done
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
