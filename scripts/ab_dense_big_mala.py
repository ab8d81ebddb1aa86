#!/usr/bin/env python3
"""MALA and MH on a dense Gaussian at D = 256, 65,536 chains: transitions/s and FP64 TFLOP/s (2 D^2 flop per transition and chain: one gradient), from the
library's HIP events.  KLARA_DENSE_NO_STREAM=1 gives the closure form.   usage: ab_dense_big_mala.py [tag]"""
import os
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import klara_jl_amd as K
from klara_jl_amd import _lib as L

tag = sys.argv[1] if len(sys.argv) > 1 else ""
n, d = 65536, int(os.environ.get("AB_D", "256"))
slow = "KLARA_DENSE_NO_STREAM" in os.environ
drift = float(os.environ.get("AB_DRIFT", "0.002"))      # (0.002: every proposal accepted; AB_DRIFT=0.012 rejects about every third)
for name, kw in (("MALA", dict(sampler=L.SAMPLER_MALA, driftstep=drift)), ("MH", dict(sampler=L.SAMPLER_MH, mh_sigma=np.full(d, 0.02)))):
    e = K.Engine(target=K.GaussDenseTarget.compound_symmetric(d, 0.5), nchains=n, nsteps=10 ** 6, steps_per_launch=8 if slow else 32, **kw)
    e.init_state_normal(); e.run(8 if slow else 64)
    r = []
    for _ in range(2 if slow else 3):
        k = 8 if slow else 128
        e.run(k); ms, nl = e.last_run_ms(); r.append(n * k / (ms * 1e-3))
    print(f"{tag} dense {name} D={d} layout {e.layout()}: transitions/s " + " ".join(f"{v:.4g}" for v in r)
          + "  TFLOP/s " + " ".join(f"{v * 2 * d * d / 1e12:.1f}" for v in r) + f"  acceptance {e.accept_counts()[0].mean() / max(e.accept_counts()[1], 1):.3f}")
    e.close()
