for i in 1 2 3; do
for lib in libklara_hip_old.so libklara_hip.so; do
for s in 1 2; do
KLARA_HIP_LIB=$GRAFT_REPO_ROOT/klara.jl_amd/lib/$lib python bench.py --streams $s --no-extra --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"$lib\", \"streams\", $s, round(d[\"ms_per_step\"]*1e3,2), round(d[\"roofline\"][\"launch_us\"],2))"
done; done; done
