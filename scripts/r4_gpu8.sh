#!/bin/bash
# round 4, GPU call 8: HMC on the dense target beyond D = 128 (streamed P, momentum in LDS): parity, then rates against the closure form
cd "$(dirname "$0")/.."
out=gpurun_out/r4_gpu8; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q -k "dense or layout_choice or random_configurations" > $out/pytest.log 2>&1
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $out/pytest.log | tail -25
timeout 600 python scripts/ab_dense_big.py stream 256 192 160 > $out/ab_dense_big.txt 2>&1
KLARA_DENSE_NO_STREAM=1 timeout 900 python scripts/ab_dense_big.py closure 256 160 >> $out/ab_dense_big.txt 2>&1
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $out/ab_dense_big.txt
