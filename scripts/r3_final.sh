#!/bin/bash
# round 3 closing pass on one box: every GPU test, the profile round (kernel traces + PMC), the bench lines, the probes
mkdir -p gpurun_out
( timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -12 ) > gpurun_out/r3_gputests.log
bash scripts/profile_round.sh r3 > gpurun_out/profile_round_r3.log 2>&1
cp gpurun_out/prof_r3/pmc_kernels.json profiles/r3_pmc_kernels.json      # (the bench lines below price with the counters of THIS box and build)
python bench.py --steps 20 --warmup 5 > gpurun_out/r3_bench_driver_flags.json 2> gpurun_out/r3_bench_driver_flags.err
python bench.py > gpurun_out/r3_bench_default.json 2> gpurun_out/r3_bench_default.err
( timeout 900 python scripts/probe_acceptance_cost.py 2>&1 ) > gpurun_out/r3_acceptance_cost_probe.txt
( timeout 900 python scripts/probe_custom_rate.py 2>&1 | tail -20 ) > gpurun_out/r3_custom_rate_probe.txt
( timeout 300 python scripts/ab_cfg45.py 2>&1 ) > gpurun_out/r3_cfg45_slice_rates.txt
( AB_CFG=slice timeout 300 python scripts/ab_cfg45.py 2>&1 ) >> gpurun_out/r3_cfg45_slice_rates.txt
tail -3 gpurun_out/r3_gputests.log; tail -2 gpurun_out/profile_round_r3.log; cut -c1-400 gpurun_out/r3_bench_driver_flags.json
