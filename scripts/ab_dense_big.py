#!/usr/bin/env python3
"""HMC (L = 10) on a dense Gaussian beyond D = 128: leapfrog*chain/s and FP64 TFLOP/s (2 D^2 + 6 D flop per leapfrog and chain) from the library's own
HIP events.  KLARA_DENSE_NO_STREAM=1 in the environment gives the closure form (one chain per lane) for comparison.   usage: ab_dense_big.py [tag] [D ...]"""
import os
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import klara_jl_amd as K
from klara_jl_amd import _lib as L

tag = sys.argv[1] if len(sys.argv) > 1 else ""
dims = [int(v) for v in sys.argv[2:]] or [256, 192, 160]
n = 65536
slow = "KLARA_DENSE_NO_STREAM" in os.environ
for d in dims:
    e = K.Engine(sampler=L.SAMPLER_HMC, target=K.GaussDenseTarget.compound_symmetric(d, 0.5), nchains=n, nsteps=10 ** 6, leapstep=0.05, nleaps=10,
                 steps_per_launch=4 if slow else 16)
    e.init_state_normal(); e.run(4 if slow else 16)
    r = []
    for _ in range(2 if slow else 3):
        k = 4 if slow else 32
        e.run(k); ms, nl = e.last_run_ms(); r.append(n * k * 10 / (ms * 1e-3))
    print(f"{tag} dense HMC L=10 D={d} layout {e.layout()}: leapfrog*chain/s " + " ".join(f"{v:.4g}" for v in r)
          + "  TFLOP/s " + " ".join(f"{v * (2 * d * d + 6 * d) / 1e12:.1f}" for v in r) + f"  acceptance {e.accept_counts()[0].mean() / max(e.accept_counts()[1], 1):.3f}")
    e.close()
