#!/usr/bin/env python3
"""Throughput of user-defined targets (run-time compiled closures, one chain per lane): the README target written as a closure and the
quartic chain of tests/cases.py, MALA, 65,536 chains."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import cases
import klara_jl_amd as K
from klara_jl_amd import _lib as L

n = 65536
for d, src, data, h in ((16, cases.SRC_NEGDOT, None, 0.1), (32, cases.SRC_NEGDOT, None, 0.1), (100, cases.SRC_NEGDOT, None, 0.05),
                        (100, cases.SRC_QUARTIC_CHAIN, [0.02, 0.5], 0.02)):
    e = K.Engine(sampler=L.SAMPLER_MALA, target=K.CustomTarget(d, src, data), nchains=n, nsteps=10 ** 6, driftstep=h)
    e.init_state_normal(); e.run(64)
    e.run(256); ms, nl = e.last_run_ms()
    print(f"custom closure D = {d:3d} ({'quartic chain' if data else 'negdot'}): {n * 256 / (ms * 1e-3):.4g} transitions/s")
    e.close()
