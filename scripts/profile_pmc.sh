#!/bin/bash
# SQ occupancy/stall counters of the bench kernel:  scripts/profile_pmc.sh <tag> [bench args]
set -u
TAG=${1:-sq}; shift || true
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 100 --warmup 10 --streams 1 --no-extra --no-cpu-baseline $*"
rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU --kernel-trace -d "$OUT/a" -o b -- python "$REPO/bench.py" $ARGS > "$OUT/a.log" 2>&1
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -d "$OUT/b" -o b -- python "$REPO/bench.py" $ARGS > "$OUT/b.log" 2>&1
python - "$OUT" <<'PY'
import csv,glob,sys,os
from collections import defaultdict
out=sys.argv[1]
for f in sorted(glob.glob(os.path.join(out,'*','**','*counter_collection.csv'),recursive=True)):
    agg=defaultdict(lambda: defaultdict(lambda:[0,0.0]))
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:60]
        if 'k_transitions' not in k and 'k_diagt<' not in k: continue
        a=agg[k][r['Counter_Name']]; a[0]+=1; a[1]+=float(r['Counter_Value'])
    for k,d in agg.items():
        print(k)
        for c,(n,v) in sorted(d.items()): print(f"   {c:24s} mean/dispatch {v/n:16.1f}")
PY
find "$OUT" -name "*.db" -delete 2>/dev/null; find "$OUT" -size +2M -delete 2>/dev/null
exit 0
