#!/usr/bin/env python3
"""Logistic regression with D = 8, 12, 16 parameters on 200 synthetic rows, MALA, 32,768 chains with running sums, for same-box A/B runs of
library builds (KLARA_HIP_LIB): transitions/s from the library's own HIP events.  (Round 3's library runs D > 8 as a one-chain-per-lane closure.)"""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import cases
import klara_jl_amd as K
from klara_jl_amd import _lib as L

tag = sys.argv[1] if len(sys.argv) > 1 else ""
n = 32768
for d in (8, 12, 16):
    X, y = cases.synthetic_logit(200, d)
    e = K.Engine(sampler=L.SAMPLER_MALA, target=K.LogisticTarget(X, y, 10.0), nchains=n, nsteps=10 ** 6, driftstep=0.02, monitor=L.MON_SUMMARIES)
    e.set_state(0.1 * np.random.default_rng(0).standard_normal((n, d))); e.run(64)
    r = []
    for _ in range(3):
        e.run(128); ms, nl = e.last_run_ms(); r.append(n * 128 / (ms * 1e-3))
    print(f"{tag} logistic MALA D={d} layout {e.layout()}: transitions/s " + " ".join(f"{v:.4g}" for v in r))
    e.close()
