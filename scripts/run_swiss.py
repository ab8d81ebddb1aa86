#!/usr/bin/env python3
"""cfg 4 share (swiss logistic MALA, 32,768 chains) — used to compare row-split factors (KLARA_LOGIT_ROWSPLIT)."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import cases
import klara_jl_amd as K
from klara_jl_amd import _lib as L

X, y = cases.swiss_data()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
x0 = np.array([5.1, -0.9, 8.2, -4.5])[None, :] + 0.1 * np.random.default_rng(0).standard_normal((n, 4))
for sampler, kw in ((L.SAMPLER_MALA, dict(driftstep=0.1)), (L.SAMPLER_HMC, dict(leapstep=0.05, nleaps=6))):
    e = K.Engine(sampler=sampler, target=K.LogisticTarget(X, y, 100.0), nchains=n, nsteps=10 ** 6, steps_per_launch=50, monitor=0, **kw)
    e.set_state(x0); e.run(100)
    t0 = time.perf_counter(); e.run(1000); dt = time.perf_counter() - t0
    print(f"swiss sampler {sampler} layout {e.layout()}: {n * 1000 / dt:.4g} transitions/s")
    e.close()
