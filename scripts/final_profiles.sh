#!/bin/bash
# Round-end refresh of profiles/: kernel stats + PMC traffic of the bench workload, kernel stats of every configuration's
# kernel, the default bench line, and the smoke check.  Everything lands under gpurun_out/ (copied into profiles/ afterwards).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 150 bash scripts/profile_bench.sh r1final2 > gpurun_out/prof_r1final2.log 2>&1
( cd /tmp && export TMPDIR=/tmp && timeout 120 rocprofv3 --output-format csv --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_r1all2" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --steps 200 --warmup 20 --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/prof_r1all2.log" 2>&1 )
find gpurun_out/prof_r1all2 -name "*.db" -delete 2>/dev/null; find gpurun_out/prof_r1all2 -size +4M -delete 2>/dev/null
timeout 120 python bench.py > gpurun_out/bench_final2.json 2> gpurun_out/bench_final2.err
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
tail -c 400 gpurun_out/bench_final2.json
