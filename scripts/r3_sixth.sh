#!/bin/bash
mkdir -p gpurun_out
L=$GRAFT_REPO_ROOT/klara.jl_amd/lib
( timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 ) > gpurun_out/r3_gputests.log
summ='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); c=d["config"]; print(sys.argv[1], "value %.3e" % d["value"], "us/step %.2f" % (d["ms_per_step"]*1e3), ["%.2f" % (v*1e3) for v in c["repetition_ms_per_step"]], c.get("launches_4lane_8lane_device_decided"), "kernel us/step", ["%.2f" % (v*1e3) for v in c.get("repetition_kernel_ms_per_step", [])], "launch_us %.1f" % d["roofline"]["launch_us"])'
for i in 1 2 3; do
for lib in libklara_hip_r2.so libklara_hip.so; do
  KLARA_HIP_LIB=$L/$lib timeout 300 python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline 2>&1 | python -c "$summ" "driver $lib"
  KLARA_HIP_LIB=$L/$lib timeout 300 python bench.py --no-extra --no-cpu-baseline 2>&1 | python -c "$summ" "default $lib"
done; done > gpurun_out/r3_ab5.log 2>&1
( timeout 900 python scripts/probe_acceptance_cost.py 2>&1 ) > gpurun_out/r3_acceptance_cost_probe.txt
cat gpurun_out/r3_gputests.log gpurun_out/r3_ab5.log; grep "spl 32" gpurun_out/r3_acceptance_cost_probe.txt
