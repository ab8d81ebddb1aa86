#!/usr/bin/env python3
"""Lanes per chain of run-time compiled whole-vector closures (KLARA_CUSTOM_LANES overrides the library's choice): user closures of O(D)
cost against the library's own closure forms of O(n D) / O(D^2) cost (logistic regression beyond 8 parameters, dense Gaussian beyond 128
dimensions), one chain per lane (the vector in registers / scratch) against 4-32 lanes with the evaluation staged through LDS."""
import os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import cases
import klara_jl_amd as K
from klara_jl_amd import _lib as L

n = 65536
for d, src, data, h in ((100, cases.SRC_NEGDOT, None, 0.05), (100, cases.SRC_QUARTIC_CHAIN, [0.02, 0.5], 0.02), (64, cases.SRC_NEGDOT, None, 0.05),
                        (256, cases.SRC_QUARTIC_CHAIN, [0.02, 0.5], 0.01)):
    for lanes in ("library", "1", "8", "16", "32"):
        os.environ.pop("KLARA_CUSTOM_LANES", None)
        if lanes != "library":
            os.environ["KLARA_CUSTOM_LANES"] = lanes
        e = K.Engine(sampler=L.SAMPLER_MALA, target=K.CustomTarget(d, src, data), nchains=n, nsteps=10 ** 6, driftstep=h, monitor=L.MON_SUMMARIES)
        e.init_state_normal(); e.run(32)
        e.run(128); ms, nl = e.last_run_ms()
        print(f"user closure D = {d:3d} ({'quartic chain' if data else 'negdot'}), lanes {lanes:>7s}: layout {e.layout()}: {n * 128 / (ms * 1e-3):.4g} transitions/s, "
              f"registers / scratch / LDS {e.kernel_attributes(0, 32)}", flush=True)
        e.close()
n = 16384
X, y = cases.synthetic_logit(400, 20)
for lanes in ("library", "4", "8", "16"):
    os.environ.pop("KLARA_CUSTOM_LANES", None)
    if lanes != "library":
        os.environ["KLARA_CUSTOM_LANES"] = lanes
    e = K.Engine(sampler=L.SAMPLER_MALA, target=K.LogisticTarget(X, y, 100.0), nchains=n, nsteps=10 ** 6, driftstep=0.01)
    e.set_state(np.zeros((n, 20))); e.run(8); e.run(32); ms, nl = e.last_run_ms()
    print(f"logistic regression D = 20, 400 rows (the library's closure form), lanes {lanes:>7s}: layout {e.layout()}: {n * 32 / (ms * 1e-3):.4g} transitions/s", flush=True)
    e.close()
for lanes in ("library", "16", "32"):
    os.environ.pop("KLARA_CUSTOM_LANES", None)
    if lanes != "library":
        os.environ["KLARA_CUSTOM_LANES"] = lanes
    e = K.Engine(sampler=L.SAMPLER_HMC, target=K.GaussDenseTarget.compound_symmetric(130, 0.5), nchains=n, nsteps=10 ** 6, leapstep=0.1, nleaps=10)
    e.init_state_normal(); e.run(2); e.run(8); ms, nl = e.last_run_ms()
    print(f"dense Gaussian D = 130, HMC L = 10 (the library's closure form), lanes {lanes:>7s}: layout {e.layout()}: {n * 8 * 10 / (ms * 1e-3):.4g} leapfrog chain/s", flush=True)
    e.close()
