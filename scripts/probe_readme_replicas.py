#!/usr/bin/env python3
"""BASELINE cfg 1 (the README job: MH, sigma = (1, 1), lt = -dot(z, z), D = 2) replicated over N chains: transitions/s from the library's
HIP events, with the running sums of mean(chain) on."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import klara_jl_amd as K
from klara_jl_amd import _lib as L

for n in (65536, 1048576):
    for mon in (L.MON_SUMMARIES, 0):
        e = K.Engine(sampler=L.SAMPLER_MH, target=K.GaussDiagTarget.negdot(2), nchains=n, nsteps=10 ** 7, burnin=1000, mh_sigma=[1.0, 1.0], monitor=mon)
        e.set_state(np.tile([5.1, -0.9], (n, 1))); e.run(1024)
        r = []
        for _ in range(3):
            e.run(2048); ms, nl = e.last_run_ms(); r.append(n * 2048 / (ms * 1e-3))
        print(f"README job x {n:8d} replicas, {'running sums' if mon else 'no monitor  '}: transitions/s " + " ".join(f"{v:.4g}" for v in r), "layout", e.layout())
        e.close()
