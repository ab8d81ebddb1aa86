#!/usr/bin/env python3
"""Rates of pair-closure jobs the pair-transposed kernels do not serve (below 17 dimensions; the slice sampler), which the library runs as whole-vector closures
(klara_custom_compose.h): chain*transitions/s from the library's HIP events, beside the same target handed over as a whole-vector closure."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import klara_jl_amd as K
from klara_jl_amd import _lib as L
import cases

n = 65536
jobs = [("pair quartic D=16 MALA", K.CustomTarget.pairwise(16, cases.SRC_PAIR_QUARTIC, [0.05, 0.3]), dict(sampler=L.SAMPLER_MALA, driftstep=0.05), 16),
        ("whole-vector quartic D=16 MALA", K.CustomTarget(16, cases.SRC_QUARTIC_CHAIN, [0.05, 0.3]), dict(sampler=L.SAMPLER_MALA, driftstep=0.05), 16),
        ("pair README closure D=8 HMC L=10", K.CustomTarget.pairwise(8, cases.SRC_PAIR_NEGDOT), dict(sampler=L.SAMPLER_HMC, leapstep=0.1, nleaps=10), 16),
        ("pair README closure D=100 slice", K.CustomTarget.pairwise(100, cases.SRC_PAIR_NEGDOT), dict(sampler=L.SAMPLER_SLICE, slice_widths=np.full(100, 1.0)), 1),
        ("whole-vector README closure D=100 slice", K.CustomTarget(100, cases.SRC_NEGDOT), dict(sampler=L.SAMPLER_SLICE, slice_widths=np.full(100, 1.0)), 1)]
for name, t, kw, spl in jobs:
    e = K.Engine(target=t, nchains=n, nsteps=10 ** 6, steps_per_launch=spl, **kw)
    e.init_state_normal(); e.run(2 * spl)
    r = []
    for _ in range(2):
        e.run(4 * spl); ms, nl = e.last_run_ms(); r.append(n * 4 * spl / (ms * 1e-3))
    print(f"{name}, {n} chains, layout {e.layout()}: chain*transitions/s " + " ".join(f"{v:.4g}" for v in r), flush=True)
    e.close()
