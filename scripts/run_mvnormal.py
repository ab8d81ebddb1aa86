#!/usr/bin/env python3
"""MALA on a non-unit diagonal Gaussian (MvNormal(mu, sigma)), D=100, 65,536 chains: unit vs general weights on layout kind 3."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import klara_jl_amd as K
from klara_jl_amd import _lib as L

n = 65536
for label, target, drift in (("negdot (unit weights)", K.GaussDiagTarget.negdot(100), 0.9),
                             ("mvnormal(mu, sigma)", K.GaussDiagTarget.mvnormal(np.linspace(-1, 1, 100), np.linspace(0.5, 1.5, 100)), 0.9)):
    for spl, steps in ((1, 400), (16, 640)):
        e = K.Engine(sampler=L.SAMPLER_MALA, target=target, nchains=n, nsteps=10 ** 6, driftstep=drift, steps_per_launch=spl, monitor=0)
        e.init_state_normal(); e.run(2 * spl); e.run(steps)
        ms, nl = e.last_run_ms()
        print(f"{label:24s} spl={spl:2d}: {ms / nl * 1e3 / spl:6.1f} us per transition of all chains ({n * steps / (ms * 1e-3):.3g} transitions/s)")
        e.close()
