#!/bin/bash
O=gpurun_out/r4_auto_threshold_atomic.txt
for i in 1 2; do
for thr in 0.03 0.05 0.08 0.12; do
KLARA_AUTO_THRESHOLD=$thr python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('threshold $thr driver flags value %.4g ms_per_step %.5f acceptance %s' % (d['value'], d['ms_per_step'], d['config'].get('acceptance_rate')))" | tee -a $O
done
done
KLARA_AUTO_THRESHOLD=0.05 python bench.py --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('threshold 0.05 default flags value %.4g ms_per_step %.5f' % (d['value'], d['ms_per_step']))" | tee -a $O
