#!/usr/bin/env python3
"""Runs only the dense-Gaussian HMC workload (BASELINE cfg 3: D=100, L=10, eps=0.1, 65,536 chains) — for profiling."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import klara_jl_amd as K
from klara_jl_amd import _lib as L

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 32
spl = int(sys.argv[3]) if len(sys.argv) > 3 else 4
e = K.Engine(sampler=L.SAMPLER_HMC, target=K.GaussDenseTarget.compound_symmetric(100, 0.5), nchains=n, nsteps=10 ** 6,
             leapstep=0.1, nleaps=10, steps_per_launch=spl)
e.init_state_normal(); e.run(2 * spl)
t0 = time.perf_counter(); e.run(steps); dt = time.perf_counter() - t0
ms, nl = e.last_run_ms()
print(f"dense HMC: {n * steps * 10 / dt:.4g} leapfrog*chain/s, {ms / nl * 1e3:.1f} us per launch of {spl} transitions, "
      f"{n * steps * 10 * 20600 / (ms * 1e-3) / 1e12:.1f} TFLOP/s algorithmic")
