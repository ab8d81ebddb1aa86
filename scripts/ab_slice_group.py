#!/usr/bin/env python3
"""The slice sampler on the group-layout kernels (every probe a full evaluation of the target by the chain's lanes): swiss logistic regression,
the rats hierarchical model, a small diagonal Gaussian and a run-time compiled closure.  chain*transitions/s from the library's HIP events.
usage: ab_slice_group.py [tag]"""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import klara_jl_amd as K
from klara_jl_amd import _lib as L
import cases

tag = sys.argv[1] if len(sys.argv) > 1 else ""
X, y = cases.swiss_data()
rats = cases.rats_target()
jobs = [("swiss logistic D=4", K.LogisticTarget(X, y, 100.0), 32768, np.full(4, 1.0), lambda n, d: 0.1 * np.random.default_rng(3).standard_normal((n, d)), 8),
        ("rats hierarchical D=%d" % rats.ndims, rats, 16384, np.full(rats.ndims, 0.5),
         lambda n, d: rats.least_squares_start()[None, :] + 0.02 * np.random.default_rng(4).standard_normal((n, d)), 2),
        ("diag Gaussian D=8", K.GaussDiagTarget.negdot(8), 65536, np.full(8, 2.0), None, 16),
        ("custom quartic D=16", K.CustomTarget(16, cases.SRC_QUARTIC_CHAIN, [0.05, 0.3]), 65536, np.full(16, 1.0), None, 4)]
for name, target, n, w, x0, spl in jobs:
    for so in (True, False):
        e = K.Engine(sampler=L.SAMPLER_SLICE, target=target, nchains=n, nsteps=10 ** 6, slice_widths=w, slice_stepout=so, steps_per_launch=spl)
        if x0 is None: e.init_state_normal()
        else: e.set_state(x0(n, target.ndims))
        e.run(2 * spl)
        r = []
        for _ in range(3):
            e.run(4 * spl); ms, nl = e.last_run_ms(); r.append(n * 4 * spl / (ms * 1e-3))
        print(f"{tag} slice {name} stepout={int(so)} chains={n} layout {e.layout()}: chain*transitions/s " + " ".join(f"{v:.4g}" for v in r))
        e.close()
