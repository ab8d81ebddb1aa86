#!/bin/bash
# Counters of the workgroup-split dense kernels (klara_dense_split.h): scripts/profile_dense_split.sh <D> [lib tag] -> gpurun_out/split_pmc_<D>_<tag>/summary.txt
# (PMC passes in their own runs with --kernel-trace only)
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
D=${1:-256}; TAG=${2:-}
OUT=$REPO/gpurun_out/split_pmc_${D}_${TAG:-default}
mkdir -p "$OUT"
[ -n "$TAG" ] && export KLARA_HIP_LIB=$REPO/klara.jl_amd/lib/libklara_hip_$TAG.so
export AB_SAMPLERS=${AB_SAMPLERS:-mala}
cd /tmp && export TMPDIR=/tmp
{
python "$REPO/scripts/ab_dense_split.py" "$TAG" $D
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d "$OUT/a" -o d -- python "$REPO/scripts/ab_dense_split.py" "$TAG" $D > "$OUT/a.log" 2>&1
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM --kernel-trace -d "$OUT/b" -o d -- python "$REPO/scripts/ab_dense_split.py" "$TAG" $D > "$OUT/b.log" 2>&1
python - "$OUT" <<'PY'
import csv,glob,sys,os
from collections import defaultdict
out=sys.argv[1]
for f in sorted(glob.glob(os.path.join(out,'*','**','*counter_collection.csv'),recursive=True)):
    agg=defaultdict(lambda:[0,0.0])
    for r in csv.DictReader(open(f)):
        if 'k_dense_split<' not in r['Kernel_Name'] and 'k_dense_big<' not in r['Kernel_Name']: continue
        a=agg[(r['Kernel_Name'][:40], r['Counter_Name'])]; a[0]+=1; a[1]+=float(r['Counter_Value'])
    for c,(n,v) in sorted(agg.items()): print(f"{c[0]:42s} {c[1]:28s} {v/n:18,.0f}   (mean per launch, n = {n})")
PY
} > "$OUT/summary.txt" 2>&1
find "$OUT" -name "*.db" -delete 2>/dev/null; find "$OUT" -size +2M -delete 2>/dev/null
cat "$OUT/summary.txt"
