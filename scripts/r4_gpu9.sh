#!/bin/bash
# round 4, GPU call 9: all four streamed dense layouts (NE = 40 / 48 / 56 / 64): parity incl. the canary run, rates at D = 160 / 224 / 256
cd "$(dirname "$0")/.."
out=gpurun_out/r4_gpu9; mkdir -p $out
timeout 1200 python -m pytest tests -m gpu -x -q -k "dense or layout_choice or canar" > $out/pytest.log 2>&1
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $out/pytest.log | tail -6
timeout 600 python scripts/ab_dense_big.py stream 256 224 192 160 130 > $out/ab_dense_big.txt 2>&1
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $out/ab_dense_big.txt
