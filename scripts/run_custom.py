#!/usr/bin/env python3
"""User-defined targets (KLARA_TARGET_CUSTOM) beside the built-in families on the same problems."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import cases
import klara_jl_amd as K
from klara_jl_amd import _lib as L


def rate(target, n, x0, steps=400, **kw):
    t0 = time.perf_counter()
    e = K.Engine(target=target, nchains=n, nsteps=10 ** 6, monitor=0, steps_per_launch=16, **kw)
    tc = time.perf_counter() - t0
    e.set_state(x0); e.run(64)
    t0 = time.perf_counter(); e.run(steps); dt = time.perf_counter() - t0
    lay = e.layout(); e.close()
    return n * steps / dt, lay, tc


n = 1 << 20
for d in (3, 10, 32, 64):
    x0 = np.random.default_rng(1).standard_normal((n, d)) * 0.5
    for name, tgt in (("built-in diag", K.GaussDiagTarget.negdot(d)), ("user source ", K.CustomTarget(d, cases.SRC_NEGDOT))):
        r, lay, tc = rate(tgt, n, x0, sampler=L.SAMPLER_MALA, driftstep=0.5)
        print(f"MALA -|x|^2 D={d:3d} {name}: {r:.3e} transitions/s  layout {lay}  create {tc:.2f} s")
X, y = cases.swiss_data()
n = 1 << 18
x0 = np.array([5.1, -0.9, 8.2, -4.5])[None, :] + 0.1 * np.random.default_rng(0).standard_normal((n, 4))
for name, tgt in (("built-in logistic", K.LogisticTarget(X, y, 100.0)),
                  ("user source     ", K.CustomTarget(4, cases.SRC_LOGIT, np.concatenate([X.ravel(), y, [100.0]])))):
    r, lay, tc = rate(tgt, n, x0, steps=200, sampler=L.SAMPLER_MALA, driftstep=0.1)
    print(f"MALA swiss (200 x 4) {name}: {r:.3e} transitions/s  layout {lay}  create {tc:.2f} s")
