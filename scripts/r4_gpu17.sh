#!/bin/bash
mkdir -p gpurun_out/r4_gpu17
for i in 1 2; do
AB_TAG=fence2 python scripts/ab_headline.py >> gpurun_out/r4_gpu17/ab.txt 2>&1
for v in 1 4; do KLARA_HIP_LIB=klara.jl_amd/lib/libklara_hip_fence$v.so AB_TAG=fence$v python scripts/ab_headline.py >> gpurun_out/r4_gpu17/ab.txt 2>&1; done
done
cat gpurun_out/r4_gpu17/ab.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "every_streamed or wide" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_workloads.py -q -k "tail" 2>&1 | tail -3
