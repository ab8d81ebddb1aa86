#!/usr/bin/env python3
"""MALA on the dense-Gaussian target (transposed 16-chains-per-wave layout), 1 and 16 transitions per launch — for profiling."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import klara_jl_amd as K
from klara_jl_amd import _lib as L

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
for spl, steps in ((1, 400), (16, 640)):
    e = K.Engine(sampler=L.SAMPLER_MALA, target=K.GaussDenseTarget.compound_symmetric(100, 0.0), nchains=n, nsteps=10 ** 6,
                 driftstep=0.9, steps_per_launch=spl)
    e.init_state_normal(); e.run(2 * spl)
    e.run(steps)
    ms, nl = e.last_run_ms()
    print(f"dense-layout MALA spl={spl}: {ms / nl * 1e3:.1f} us per launch, {n * steps / (ms * 1e-3):.4g} transitions/s")
    e.close()
