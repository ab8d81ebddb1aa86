#!/bin/bash
mkdir -p gpurun_out/r4_gpu16
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/r4_gpu16/pytest.log
cat gpurun_out/r4_gpu16/pytest.log
