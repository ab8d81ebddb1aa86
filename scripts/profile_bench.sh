#!/bin/bash
# Profiles the bench.py workload with rocprofv3 on the GPU box and leaves summaries in gpurun_out/prof_<tag>/.
#   scripts/profile_bench.sh <tag> [bench args...]
# Pass 1: --kernel-trace --stats (per-kernel durations).  Passes 2,3: PMC counters FETCH_SIZE / WRITE_SIZE in
# their own runs (never combined with sys/hip/hsa tracing — see the task notes).
set -u
TAG=${1:-r1}; shift || true
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 200 --warmup 20 --streams 1 --no-extra --no-cpu-baseline $*"   # one launch at a time: kernel durations = roofline.launch_us
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/trace" -o bench -- python "$REPO/bench.py" $ARGS > "$OUT/bench_trace.log" 2>&1
rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d "$OUT/pmc_fetch" -o bench -- python "$REPO/bench.py" $ARGS > "$OUT/bench_fetch.log" 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE --kernel-trace -d "$OUT/pmc_write" -o bench -- python "$REPO/bench.py" $ARGS > "$OUT/bench_write.log" 2>&1
python "$REPO/scripts/summarize_profile.py" "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
# keep only small files for the merge back
find "$OUT" -name "*.db" -delete 2>/dev/null; ls -R "$OUT" | head -40
find "$OUT" -size +4M -delete 2>/dev/null
exit 0
