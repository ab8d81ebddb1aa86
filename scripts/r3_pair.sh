#!/bin/bash
mkdir -p gpurun_out
( timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -12 ) > gpurun_out/r3_gputests.log
( timeout 900 python scripts/probe_custom_rate.py 2>&1 | tail -14 ) > gpurun_out/r3_custom_rate_probe.txt
tail -6 gpurun_out/r3_gputests.log; cat gpurun_out/r3_custom_rate_probe.txt
