#!/bin/bash
# round 4, GPU call 5: the full parity suite (4-lane row split, 9..16-parameter logistic on the row-split kernels, the canary sweeps), then the
# logistic regression at D = 8 / 12 / 16 against round 3's forms, and the cfg 4 / dense numbers of this build
cd "$(dirname "$0")/.."
out=gpurun_out/r4_gpu5; mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $out/pytest.log | tail -12
L=klara.jl_amd/lib
for v in r4base main e8w2; do
  lib=$L/libklara_hip_$v.so; [ $v = main ] && lib=$L/libklara_hip.so
  KLARA_HIP_LIB=$lib timeout 600 python scripts/ab_logit_wide.py $v >> $out/ab_logit_wide.txt 2>&1
done
for v in r4base main; do
  lib=$L/libklara_hip_$v.so; [ $v = main ] && lib=$L/libklara_hip.so
  KLARA_HIP_LIB=$lib timeout 300 python scripts/ab_logit.py $v >> $out/ab_logit.txt 2>&1
done
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $out/ab_logit_wide.txt $out/ab_logit.txt
