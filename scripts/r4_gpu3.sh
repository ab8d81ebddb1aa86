#!/bin/bash
# round 4, GPU call 3: full parity suite on the new logistic-row functions (kd_exp_neg with the one-fma reduction, kd_log12), then same-box
# A/B of waves per SIMD x rows per batch x row split against the build before this round's logistic work, MALA (cfg 4) and HMC
cd "$(dirname "$0")/.."
out=gpurun_out/r4_gpu3; mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1
tail -3 $out/pytest.log
L=klara.jl_amd/lib
for v in r4base main w2b5 w2b6 w3b3; do
  lib=$L/libklara_hip_$v.so; [ $v = main ] && lib=$L/libklara_hip.so
  for rs in 8 4; do
    KLARA_HIP_LIB=$lib KLARA_LOGIT_ROWSPLIT=$rs timeout 300 python scripts/ab_logit.py "$v rs$rs" >> $out/ab_logit.txt 2>&1
  done
done
for v in r4base main w3b3; do
  lib=$L/libklara_hip_$v.so; [ $v = main ] && lib=$L/libklara_hip.so
  for rs in 8 4; do
    KLARA_HIP_LIB=$lib KLARA_LOGIT_ROWSPLIT=$rs timeout 300 python scripts/ab_logit_hmc.py "$v rs$rs" >> $out/ab_logit_hmc.txt 2>&1
  done
done
cat $out/ab_logit.txt $out/ab_logit_hmc.txt
