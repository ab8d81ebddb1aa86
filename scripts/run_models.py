#!/usr/bin/env python3
"""Per-GPU throughput of the two data-model configs of BASELINE.json (cfg 4 swiss logistic MALA, cfg 5 rats HMC L=32)."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import cases
import klara_jl_amd as K
from klara_jl_amd import _lib as L

X, y = cases.swiss_data()
n = 32768                                   # 262,144 chains / 8 GPUs
x0 = np.array([5.1, -0.9, 8.2, -4.5])[None, :] + 0.1 * np.random.default_rng(0).standard_normal((n, 4))
e = K.Engine(sampler=L.SAMPLER_MALA, target=K.LogisticTarget(X, y, 100.0), nchains=n, nsteps=10 ** 6, driftstep=0.1,
             steps_per_launch=50)
e.set_state(x0); e.run(100)
t0 = time.perf_counter(); e.run(1000); dt = time.perf_counter() - t0
print(f"cfg 4 swiss MALA  : {n} chains, {n * 1000 / dt:.4g} transitions/s per GPU ({dt * 1e3 / 1000:.3f} ms per transition of all chains)")
e.close()

t = cases.rats_target()
n = 131072                                  # 1,048,576 chains / 8 GPUs
x0 = t.least_squares_start()[None, :] + 0.05 * np.random.default_rng(1).standard_normal((n, t.ndims))
e = K.Engine(sampler=L.SAMPLER_HMC, target=t, nchains=n, nsteps=2000, burnin=1000, leapstep=0.02, nleaps=32,
             tuner=L.TUNER_ACCEPT_RATE, tuner_mode=L.TUNE_POOLED, targetrate=0.65, period=100, steps_per_launch=10,
             monitor=L.MON_SUMMARIES)
e.set_state(x0); e.run(100)
t0 = time.perf_counter(); e.run(200); dt = time.perf_counter() - t0
print(f"cfg 5 rats HMC L=32: {n} chains, {n * 200 / dt:.4g} transitions/s = {n * 200 * 32 / dt:.4g} leapfrog*chain/s per GPU; "
      f"2000 steps would take {dt / 200 * 2000:.1f} s; pooled step {e.tune()[0][0]:.4f}")
e.close()
