#!/bin/bash
# round 4, GPU call 10: MALA / MH on the streamed dense layouts: parity (dense cases, layout mirror, random configurations, canaries), then rates
cd "$(dirname "$0")/.."
out=gpurun_out/r4_gpu10; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q -k "dense or layout_choice or canar or c_example" > $out/pytest.log 2>&1
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $out/pytest.log | tail -8
timeout 600 python scripts/ab_dense_big_mala.py stream > $out/ab_dense_big_mala.txt 2>&1
KLARA_DENSE_NO_STREAM=1 timeout 600 python scripts/ab_dense_big_mala.py closure >> $out/ab_dense_big_mala.txt 2>&1
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $out/ab_dense_big_mala.txt
