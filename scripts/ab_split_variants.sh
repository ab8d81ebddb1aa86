# same-box A/B of builds of klara_dense_split (scripts/build_variant.sh <tag> -D...; KLARA_VARIANT_TUS=klara_dense_split): rates of the split layout
# usage: AB_VARIANTS="nores ''" AB_DIMS="320 512 1024" bash scripts/ab_split_variants.sh
for rep in 1 2; do
for v in ${AB_VARIANTS:-nores default}; do
  [ "$v" = default ] && lib=klara.jl_amd/lib/libklara_hip.so || lib=klara.jl_amd/lib/libklara_hip_$v.so
  KLARA_HIP_LIB=$PWD/$lib python scripts/ab_dense_split.py "$v" ${AB_DIMS:-320 512 1024} 2>&1 | tail -9
done; done
