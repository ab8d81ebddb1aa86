#!/bin/bash
O=gpurun_out/r4_auto_threshold_atomic.txt
for i in 1 2; do
for thr in 0.055 0.06 0.065 0.07; do
KLARA_AUTO_THRESHOLD=$thr python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('threshold $thr driver flags value %.4g ms_per_step %.5f reps %s' % (d['value'], d['ms_per_step'], [round(v*1e3,2) for v in d['config'].get('reps_ms_per_step', [])] or d.get('reps')))" | tee -a $O
done
done
