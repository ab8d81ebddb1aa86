#!/bin/bash
# Round profile of the bench command on the GPU box -> gpurun_out/prof_<tag>/ (copied into profiles/ afterwards).
#   scripts/profile_round.sh <tag>
# Pass 0a/0b: rocprofv3 --kernel-trace --stats of the headline kernel alone (one launch shape) and of
# `bench.py --steps 256 --warmup 32 --reps 2 --no-cpu-baseline` with every configuration (per-kernel durations).
# Passes 1-4: PMC counters, each in its own run with --kernel-trace only (never combined with sys/hip/hsa tracing):
#   sq    SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
#   mem   SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SMEM
#   fetch FETCH_SIZE        write WRITE_SIZE
set -u
TAG=${1:-r3}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
TRACE_ARGS="--steps 256 --warmup 32 --reps 2 --no-cpu-baseline"
PMC_ARGS="--steps 64 --warmup 16 --reps 1 --no-cpu-baseline"
# Pass 0a: the headline kernel in ONE launch shape (every launch of the whole job on one stream, no other configuration): its
# average duration in this summary is what bench.py's roofline.launch_us must agree with
timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/trace_headline" -o bench -- python "$REPO/bench.py" --steps 512 --warmup 64 --reps 2 --streams 1 --no-extra --no-cpu-baseline > "$OUT/bench_trace_headline.log" 2>&1
f=$(find "$OUT/trace_headline" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats_headline.csv"
grep '^{"metric"' "$OUT/bench_trace_headline.log" | tail -1 > "$OUT/bench_line_trace_headline.json"
timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/trace" -o bench -- python "$REPO/bench.py" $TRACE_ARGS > "$OUT/bench_trace.log" 2>&1
timeout 300 rocprofv3 --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -d "$OUT/pmc_sq" -o bench -- python "$REPO/bench.py" $PMC_ARGS > "$OUT/bench_sq.log" 2>&1
timeout 300 rocprofv3 --output-format csv --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SMEM --kernel-trace -d "$OUT/pmc_mem" -o bench -- python "$REPO/bench.py" $PMC_ARGS > "$OUT/bench_mem.log" 2>&1
timeout 300 rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d "$OUT/pmc_fetch" -o bench -- python "$REPO/bench.py" $PMC_ARGS > "$OUT/bench_fetch.log" 2>&1
timeout 300 rocprofv3 --output-format csv --pmc WRITE_SIZE --kernel-trace -d "$OUT/pmc_write" -o bench -- python "$REPO/bench.py" $PMC_ARGS > "$OUT/bench_write.log" 2>&1
for l in trace sq mem fetch write; do grep '^{"metric"' "$OUT/bench_$l.log" | tail -1 > "$OUT/bench_line_$l.json"; done
python "$REPO/scripts/make_pmc_json.py" "$OUT" "$OUT/pmc_kernels.json" > "$OUT/pmc_summary.txt" 2>&1
f=$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv"
# keep only small files for the merge back
find "$OUT" -name "*.db" -delete 2>/dev/null
find "$OUT" -name "*kernel_trace.csv" -size +2M -delete 2>/dev/null
find "$OUT" -name "*counter_collection.csv" -delete 2>/dev/null
find "$OUT" -size +4M -delete 2>/dev/null
head -60 "$OUT/pmc_summary.txt"
exit 0
