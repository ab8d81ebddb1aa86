#!/usr/bin/env python3
"""MALA on lt = -|x|^2 around the D = 128 boundary between the pair-transposed layout (kind 3) and the group layout (kind 0)."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import klara_jl_amd as K
from klara_jl_amd import _lib as L

n = 65536
for d in (100, 128, 130, 192, 256, 384, 512):
    for spl in (1, 16):
        e = K.Engine(sampler=L.SAMPLER_MALA, target=K.GaussDiagTarget.negdot(d), nchains=n, nsteps=10 ** 6, driftstep=0.5, monitor=0,
                     steps_per_launch=spl)
        e.init_state_normal(); e.run(64)
        t0 = time.perf_counter(); e.run(320); dt = time.perf_counter() - t0
        print(f"D={d:4d} spl={spl:2d} layout {e.layout()}: {n * 320 / dt:.3e} transitions/s = {n * 320 * d / dt:.3e} element-transitions/s")
        e.close()
