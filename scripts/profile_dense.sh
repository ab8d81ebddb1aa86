#!/bin/bash
# MFMA counters of the dense-target HMC kernel (BASELINE cfg 3) + throughput at several transitions per launch.
#   scripts/profile_dense.sh   -> gpurun_out/dense_pmc/summary.txt   (PMC passes never combined with sys/hip tracing)
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/dense_pmc
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
{
echo "# rocprofv3 --pmc (two passes, --kernel-trace only) on scripts/run_dense.py 65536 16 4"
python "$REPO/scripts/run_dense.py" 65536 16 4
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d "$OUT/a" -o d -- python "$REPO/scripts/run_dense.py" 65536 16 4 > "$OUT/a.log" 2>&1
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY --kernel-trace -d "$OUT/b" -o d -- python "$REPO/scripts/run_dense.py" 65536 16 4 > "$OUT/b.log" 2>&1
python - "$OUT" <<'PY'
import csv,glob,sys,os
from collections import defaultdict
out=sys.argv[1]
for f in sorted(glob.glob(os.path.join(out,'*','**','*counter_collection.csv'),recursive=True)):
    agg=defaultdict(lambda:[0,0.0])
    for r in csv.DictReader(open(f)):
        if 'k_dense_transitions' not in r['Kernel_Name']: continue
        a=agg[r['Counter_Name']]; a[0]+=1; a[1]+=float(r['Counter_Value'])
    for c,(n,v) in sorted(agg.items()): print(f"{c:28s} {v/n:18,.0f}   (mean per launch, n = {n})")
PY
echo "# throughput vs transitions per launch"
for S in 1 4 16 64; do python "$REPO/scripts/run_dense.py" 65536 $((S*4 > 64 ? S*4 : 64)) $S; done
} > "$OUT/summary.txt" 2>&1
find "$OUT" -name "*.db" -delete 2>/dev/null; find "$OUT" -size +2M -delete 2>/dev/null
cat "$OUT/summary.txt"
