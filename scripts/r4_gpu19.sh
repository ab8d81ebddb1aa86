#!/bin/bash
mkdir -p gpurun_out/r4_gpu19
for spl in 32 64 128 256 512; do
python bench.py --no-extra --no-cpu-baseline --spl $spl 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('spl $spl value %.4g ms_per_step %.5f launch_us %.1f' % (d['value'], d['ms_per_step'], d['roofline']['launch_us']))" | tee -a gpurun_out/r4_gpu19/spl.txt
done
