#!/bin/bash
# round 4, GPU call 4: the canary tests; then the driver's 20-transition region issued as chain partitions x short launches (--streams x --spl)
cd "$(dirname "$0")/.."
out=gpurun_out/r4_gpu4; mkdir -p $out
python -m pytest tests/test_gpu_canary.py -x -q > $out/pytest_canary.log 2>&1
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $out/pytest_canary.log | tail -15
run() { python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$*', 'value %.4g ms_per_step %.5f' % (d['value'], d['ms_per_step']), 'reps', [round(v,5) for v in d['config'].get('repetition_ms_per_step',[])])"; }
for rep in 1 2; do
  run
  for st in 2 4; do for spl in 4 5 7 10; do run --streams $st --spl $spl; done; done
done > $out/short_runs.txt 2>&1
for st in 0 3 4; do python bench.py --no-extra --no-cpu-baseline --streams $st 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('default flags streams $st', 'value %.4g ms_per_step %.5f' % (d['value'], d['ms_per_step']))"; done >> $out/short_runs.txt 2>&1
cat $out/short_runs.txt
