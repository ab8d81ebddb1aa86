#!/usr/bin/env python3
"""Slice sampler (SliceSampler.jl:60-109) on the README target: transitions/s and coordinate updates/s."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import klara_jl_amd as K
from klara_jl_amd import _lib as L

for d, n, steps in ((100, 65536, 8), (10, 65536, 60), (4, 65536, 200)):
    e = K.Engine(sampler=L.SAMPLER_SLICE, target=K.GaussDiagTarget.negdot(d), nchains=n, nsteps=10 ** 6, slice_widths=np.full(d, 1.0),
                 steps_per_launch=4, monitor=0)
    e.init_state_normal(); e.run(4)
    t0 = time.perf_counter(); e.run(steps); dt = time.perf_counter() - t0
    print(f"slice D={d} layout {e.layout()}: {n * steps / dt:.4g} transitions/s = {n * steps * d / dt:.4g} coordinate updates/s")
    e.close()
