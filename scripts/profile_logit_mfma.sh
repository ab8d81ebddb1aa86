#!/bin/bash
# Counters of the matrix-core logistic kernels (k_logit_mfma, klara_logit_mfma.h): two rocprofv3 --pmc passes with --kernel-trace only on
# scripts/probe_logit_dims.py (MALA and HMC at 64 x 200 and 128 x 1000).   scripts/profile_logit_mfma.sh -> gpurun_out/logit_mfma_pmc/summary.txt
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/logit_mfma_pmc
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export AB_SAMPLERS=MALA,HMC AB_SHAPES=${AB_SHAPES:-32x200,64x200,128x1000}
{
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES --kernel-trace -d "$OUT/p1" -o d -- python "$REPO/scripts/probe_logit_dims.py" > "$OUT/p1.log" 2>&1
rocprofv3 --output-format csv --pmc SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES --kernel-trace -d "$OUT/p2" -o d -- python "$REPO/scripts/probe_logit_dims.py" > "$OUT/p2.log" 2>&1
grep logistic "$OUT/p1.log"
python - "$OUT" <<'PY'
import csv,glob,sys,os
from collections import defaultdict
out=sys.argv[1]
agg=defaultdict(lambda: defaultdict(list)); dur=defaultdict(list)
for p in ("p1","p2"):
    for f in glob.glob(os.path.join(out,p,'**','*counter_collection.csv'),recursive=True):
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'].split('(')[0]
            if 'k_logit_mfma<' in k: agg[(k,r.get('Grid_Size',''))][r['Counter_Name']].append(float(r['Counter_Value']))
    for f in glob.glob(os.path.join(out,p,'**','*kernel_trace.csv'),recursive=True):
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'].split('(')[0]
            if 'k_logit_mfma<' in k: dur[k].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))*1e-3)
for (k,g),c in sorted(agg.items()):
    m=lambda n: sum(c[n])/max(len(c[n]),1) if n in c else float('nan')
    us=sum(dur[k])/max(len(dur[k]),1)
    mf=m('SQ_INSTS_VALU_MFMA_F64'); valu=m('SQ_INSTS_VALU'); gui=m('GRBM_GUI_ACTIVE')
    print(f"{k} grid {g}: {len(c['SQ_INSTS_VALU'])} launches, {us:,.0f} us each (mean over this kernel name under the counters)")
    print(f"   MFMA {mf:,.0f}, other vector {valu-mf:,.0f} instructions per launch; MFMA pipes busy {m('SQ_VALU_MFMA_BUSY_CYCLES')/(gui*4*32):.2f} of the launch; "
          f"VALU-issue busy (4 x SQ_ACTIVE_INST_VALU / (GUI cycles x 1024 SIMDs / 8 XCDs ...)) {4*m('SQ_ACTIVE_INST_VALU')/(gui/8*1024) if gui==gui else float('nan'):.2f}; "
          f"SQ_WAIT_ANY / SQ_WAVE_CYCLES {m('SQ_WAIT_ANY')/m('SQ_WAVE_CYCLES'):.2f}; LDS instr {m('SQ_INSTS_LDS'):,.0f}, VMEM reads {m('SQ_INSTS_VMEM_RD'):,.0f}")
PY
} 2>&1 | tee "$OUT/summary.txt"
find "$OUT" -name "*.db" -delete 2>/dev/null; find "$OUT" -name "*counter_collection.csv" -delete 2>/dev/null; find "$OUT" -name "*kernel_trace.csv" -size +2M -delete 2>/dev/null
exit 0
