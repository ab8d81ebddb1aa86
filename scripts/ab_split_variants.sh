# same-box A/B of builds of klara_dense_split (scripts/build_variant.sh): rates of the split layout at D = 256 (KLARA_DENSE_SPLIT=1), 512, 1024
for rep in 1 2; do
for v in ${AB_VARIANTS:-r8 "" r32}; do
  lib=klara.jl_amd/lib/libklara_hip${v:+_$v}.so
  KLARA_HIP_LIB=$PWD/$lib python scripts/ab_dense_split.py "${v:-default}" 512 1024 2>&1 | tail -6
  KLARA_DENSE_SPLIT=1 KLARA_HIP_LIB=$PWD/$lib python scripts/ab_dense_split.py "${v:-default}" 256 2>&1 | tail -3
done; done
