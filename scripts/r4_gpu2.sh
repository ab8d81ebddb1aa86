#!/bin/bash
# round 4, GPU call 2: parity of the staged logistic rows and of the dense kernel's register-resident state, then same-box A/B of the
# logistic variants (waves per SIMD x rows per batch x row split) and of the dense kernel against the build before these changes
cd "$(dirname "$0")/.."
out=gpurun_out/r4_gpu2; mkdir -p $out
python -m pytest tests -m gpu -x -q -k "swiss or logit or dense or cfg3 or cfg4 or workloads or random_configurations" > $out/pytest.log 2>&1
tail -3 $out/pytest.log
L=klara.jl_amd/lib
for v in r4base main w2b5 w2b3 w3b3 w4b2; do
  lib=$L/libklara_hip_$v.so; [ $v = main ] && lib=$L/libklara_hip.so
  for rs in 8 4; do
    KLARA_HIP_LIB=$lib KLARA_LOGIT_ROWSPLIT=$rs timeout 300 python scripts/ab_logit.py "$v rs$rs" >> $out/ab_logit.txt 2>&1
  done
done
for v in r4base main r4base main; do
  lib=$L/libklara_hip_$v.so; [ $v = main ] && lib=$L/libklara_hip.so
  KLARA_HIP_LIB=$lib timeout 300 python scripts/ab_models.py $v 2>&1 | grep cfg3 >> $out/ab_dense.txt
done
cat $out/ab_logit.txt $out/ab_dense.txt
