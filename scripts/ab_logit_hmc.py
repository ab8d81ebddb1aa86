#!/usr/bin/env python3
"""HMC (leapstep 0.05, 10 leapfrogs) on the swiss logistic regression, 32,768 chains, for same-box A/B runs of library builds (KLARA_HIP_LIB)."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import cases
import klara_jl_amd as K
from klara_jl_amd import _lib as L

tag = sys.argv[1] if len(sys.argv) > 1 else ""
X, y = cases.swiss_data()
n = 32768
x0 = np.array([5.1, -0.9, 8.2, -4.5])[None, :] + 0.1 * np.random.default_rng(0).standard_normal((n, 4))
for mon in (L.MON_SUMMARIES, 0):
    e = K.Engine(sampler=L.SAMPLER_HMC, target=K.LogisticTarget(X, y, 100.0), nchains=n, nsteps=10 ** 6, leapstep=0.05, nleaps=10, monitor=mon)
    e.set_state(x0); e.run(64)
    r = []
    for _ in range(3):
        e.run(128); ms, nl = e.last_run_ms(); r.append(n * 128 * 10 / (ms * 1e-3))
    print(f"{tag} swiss HMC L=10 {'with sums' if mon else 'no monitor'}: leapfrog*chain/s " + " ".join(f"{v:.4g}" for v in r))
    e.close()
