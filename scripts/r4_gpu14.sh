#!/bin/bash
mkdir -p gpurun_out/r4_gpu14
for i in 1 2; do
AB_TAG=base python scripts/ab_headline.py >> gpurun_out/r4_gpu14/ab.txt 2>&1
KLARA_HIP_LIB=klara.jl_amd/lib/libklara_hip_share.so AB_TAG=share python scripts/ab_headline.py >> gpurun_out/r4_gpu14/ab.txt 2>&1
done
cat gpurun_out/r4_gpu14/ab.txt
