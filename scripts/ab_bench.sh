#!/bin/bash
# Same-box A/B of two builds of the library (box-to-box variation on the pool is +-3 %, more than most kernel changes):
#   make -C klara.jl_amd/csrc OUT=../lib/libklara_hip_old.so OBJDIR=../../build/csrc_old   (at the older commit)
#   gpurun -- 'bash scripts/ab_bench.sh'
# prints ms per transition of all chains (bench value) and the single-launch duration for 1 and 2 streams, alternating.
for i in 1 2 3; do
for lib in libklara_hip_old.so libklara_hip.so; do
for s in 1 2; do
KLARA_HIP_LIB=$GRAFT_REPO_ROOT/klara.jl_amd/lib/$lib python bench.py --streams $s --no-extra --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"$lib\", \"streams\", $s, round(d[\"ms_per_step\"]*1e3,2), round(d[\"roofline\"][\"launch_us\"],2))"
done; done; done
