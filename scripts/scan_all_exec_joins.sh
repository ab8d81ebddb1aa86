#!/bin/bash
# scan_exec_joins.py over every translation unit of the library (and the 4 / 16 / 32-lane builds of the pair-transposed ones): prints the kernels with a hit
cd "$(dirname "$0")/.."
for tu in klara_mh klara_mala klara_hmc klara_slice klara_dense klara_dense_big klara_hiert klara_diagt_init; do
  echo "== $tu: $(python scripts/scan_exec_joins.py $tu 2>&1 | awk '$1 ~ /^[0-9]+$/ {n++; if ($1 > 0) h++} END {printf "%d kernels, %d with a hit", n, h+0}')"
done
for tu in klara_diagt_mh klara_diagt_mala klara_diagt_hmc klara_diagt_slice; do
  for q in 8 4 16 32; do
    case "$tu-$q" in klara_diagt_hmc-4|klara_diagt_slice-4) continue;; esac
    echo "== $tu Q=$q: $(python scripts/scan_exec_joins.py $tu -DKLARA_DIAGT_Q=$q 2>&1 | awk '$1 ~ /^[0-9]+$/ {n++; if ($1 > 0) h++} END {printf "%d kernels, %d with a hit", n, h+0}')"
  done
done
