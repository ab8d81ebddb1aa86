#!/bin/bash
mkdir -p gpurun_out/r4_gpu22
O=gpurun_out/r4_gpu22/ab_fold.txt
for i in 1 2; do
for k in 0 1 2 16; do
lib=klara.jl_amd/lib/libklara_hip_fa$k.so; [ $k = 1 ] && lib=klara.jl_amd/lib/libklara_hip.so
KLARA_HIP_LIB=$lib AB_TAG=atomic_max_$k python scripts/ab_headline.py 2>&1 | tail -1 | tee -a $O
KLARA_HIP_LIB=$lib python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('atomic_max_$k driver flags value %.4g ms_per_step %.5f' % (d['value'], d['ms_per_step']))" | tee -a $O
done
done
