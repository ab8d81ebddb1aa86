#!/usr/bin/env python3
"""Slice sampler on the README target, D = 100, 65,536 chains: coordinate updates/s from the library's HIP events (KLARA_HIP_LIB selects the build)."""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import klara_jl_amd as K
from klara_jl_amd import _lib as L
n, d = 65536, 100
for mon in (0, L.MON_SUMMARIES):
    e = K.Engine(sampler=L.SAMPLER_SLICE, target=K.GaussDiagTarget.negdot(d), nchains=n, nsteps=10 ** 7, slice_widths=np.full(d, 1.0), steps_per_launch=32, monitor=mon, nstreams=1)
    e.init_state_normal(); e.run(64)
    r = []
    for _ in range(3):
        e.run(128); ms, nl = e.last_run_ms(); r.append(n * 128 * d / (ms * 1e-3))
    print(f"[{os.path.basename(os.environ.get('KLARA_HIP_LIB', 'default'))}] slice D={d} {'running sums' if mon else 'no monitor'}: coordinate updates/s " + " ".join(f"{v:.4g}" for v in r), flush=True)
    e.close()
