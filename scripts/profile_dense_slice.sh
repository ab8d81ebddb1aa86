#!/bin/bash
# MFMA counters of the slice sampler's dense-target kernels (k_dense_transitions<SLICE, 25>, k_dense_big<SLICE, 64>): one rocprofv3 --pmc pass with
# --kernel-trace only, on scripts/ab_dense_slice.py.   scripts/profile_dense_slice.sh -> gpurun_out/dense_slice_pmc/summary.txt
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/dense_slice_pmc
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
{
for cfg in "65536 100" "16384 256"; do
  set -- $cfg
  rocprofv3 --output-format csv --pmc SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES --kernel-trace -d "$OUT/d$2" -o d -- python "$REPO/scripts/ab_dense_slice.py" pmc $1 $2 > "$OUT/d$2.log" 2>&1
  grep "dense slice" "$OUT/d$2.log"
done
python - "$OUT" <<'PY'
import csv,glob,sys,os
from collections import defaultdict
out=sys.argv[1]
for d in sorted(glob.glob(os.path.join(out,'d*'))):
    if not os.path.isdir(d): continue
    cc=glob.glob(os.path.join(d,'**','*counter_collection.csv'),recursive=True)
    kt=glob.glob(os.path.join(d,'**','*kernel_trace.csv'),recursive=True)
    dur=defaultdict(list)
    for f in kt:
        for r in csv.DictReader(open(f)):
            if 'SLICE' in r['Kernel_Name'] or 'Li3E' in r['Kernel_Name'] or '<3,' in r['Kernel_Name']:
                dur[r['Kernel_Name'].split('(')[0]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))*1e-3)
    agg=defaultdict(lambda: defaultdict(list))
    for f in cc:
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'].split('(')[0]
            if '<3,' not in k: continue
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,c in agg.items():
        print(os.path.basename(d), k)
        n=len(c['SQ_INSTS_VALU_MFMA_F64'])
        mf=sum(c['SQ_INSTS_VALU_MFMA_F64'])/n; busy=sum(c['SQ_VALU_MFMA_BUSY_CYCLES'])/n; gui=sum(c['GRBM_GUI_ACTIVE'])/n; valu=sum(c['SQ_INSTS_VALU'])/n
        us=sum(dur[k])/max(len(dur[k]),1)
        print(f"   launches {n}: MFMA instructions {mf:,.0f}, other vector instructions {valu-mf:,.0f} per launch; kernel {us:,.0f} us (under the counters)")
        print(f"   MFMA work {mf*2048/1e12:.3f} Tflop per launch -> {mf*2048/(us*1e-6)/1e12:.1f} TFLOP/s executed = {mf*2048/(us*1e-6)/78.6e12:.2f} of the FP64 MFMA peak;  SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 4 SIMDs x 32 CUs) = {busy/(gui*4*32):.2f}")
PY
} > "$OUT/summary.txt" 2>&1
find "$OUT" -name "*.db" -delete 2>/dev/null; find "$OUT" -size +2M -delete 2>/dev/null
cat "$OUT/summary.txt"
