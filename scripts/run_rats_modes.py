#!/usr/bin/env python3
"""cfg 5 (rats HMC L=32) per-GPU share under different tuner settings — how much the general (MODE 0) kernel costs."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import cases
import klara_jl_amd as K
from klara_jl_amd import _lib as L

t = cases.rats_target()
n = 131072
x0 = t.least_squares_start()[None, :] + 0.05 * np.random.default_rng(1).standard_normal((n, t.ndims))
for label, kw in (("pooled AcceptanceRate + summaries", dict(tuner=L.TUNER_ACCEPT_RATE, tuner_mode=L.TUNE_POOLED, targetrate=0.65, period=100, monitor=L.MON_SUMMARIES)),
                  ("pooled AcceptanceRate, no monitor", dict(tuner=L.TUNER_ACCEPT_RATE, tuner_mode=L.TUNE_POOLED, targetrate=0.65, period=100, monitor=0)),
                  ("per-chain AcceptanceRate, no monitor", dict(tuner=L.TUNER_ACCEPT_RATE, targetrate=0.65, period=100, monitor=0)),
                  ("Vanilla + summaries", dict(monitor=L.MON_SUMMARIES)),
                  ("Vanilla, no monitor", dict(monitor=0))):
    e = K.Engine(sampler=L.SAMPLER_HMC, target=t, nchains=n, nsteps=2000, burnin=1000, leapstep=0.02, nleaps=32, steps_per_launch=10, **kw)
    e.set_state(x0); e.run(100)
    t0 = time.perf_counter(); e.run(200); dt = time.perf_counter() - t0
    print(f"{label:40s} {n * 200 * 32 / dt:.4g} leapfrog*chain/s")
    e.close()
