#!/usr/bin/env python3
"""The slice sampler on a dense Gaussian beyond D = 128 (a probe = one full evaluation of the log-target): chain*transitions/s and coordinate updates/s
from the library's HIP events.  KLARA_DENSE_SLICE_NO_STREAM=1 gives the closure form (one chain per lane).   usage: ab_dense_slice.py [tag] [nchains] [D ...]"""
import os
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import klara_jl_amd as K
from klara_jl_amd import _lib as L

tag = sys.argv[1] if len(sys.argv) > 1 else ""
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
dims = [int(v) for v in sys.argv[3:]] or [256, 160]
for d in dims:
    for so in (True, False):
        e = K.Engine(sampler=L.SAMPLER_SLICE, target=K.GaussDenseTarget.compound_symmetric(d, 0.5), nchains=n, nsteps=10 ** 6,
                     slice_widths=np.full(d, 2.0), slice_stepout=so, steps_per_launch=1)
        e.init_state_normal(); e.run(1)
        r = []
        for _ in range(2):
            e.run(1); ms, nl = e.last_run_ms(); r.append(n / (ms * 1e-3))
        print(f"{tag} dense slice D={d} stepout={int(so)} chains={n} layout {e.layout()}: chain*transitions/s " + " ".join(f"{v:.4g}" for v in r)
              + "  coordinate updates/s " + " ".join(f"{v * d:.4g}" for v in r))
        e.close()
