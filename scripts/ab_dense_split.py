#!/usr/bin/env python3
"""MALA, MH and HMC (L = 10) on a dense Gaussian, 65,536 chains, on the workgroup-split layout (klara_dense_split.h, kind 6: D > 256, or every D under
KLARA_DENSE_SPLIT=1) and on the streamed one (kind 1): rates and FP64 TFLOP/s (2 D^2 flop per gradient and chain) from the library's HIP events.
usage: ab_dense_split.py [tag] [D ...]     (AB_N chains, AB_SAMPLERS=mala,mh,hmc)"""
import os
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import klara_jl_amd as K
from klara_jl_amd import _lib as L

tag = sys.argv[1] if len(sys.argv) > 1 else ""
dims = [int(v) for v in sys.argv[2:]] or [256, 512, 1024]
n = int(os.environ.get("AB_N", "65536"))
which = os.environ.get("AB_SAMPLERS", "mala,mh,hmc").split(",")
for d in dims:
    jobs = []
    if "mala" in which: jobs.append(("MALA", 1, dict(sampler=L.SAMPLER_MALA, driftstep=float(os.environ.get("AB_DRIFT", "0.002")) * 256 / d)))
    if "mh" in which: jobs.append(("MH", 1, dict(sampler=L.SAMPLER_MH, mh_sigma=np.full(d, 0.02 * (256 / d) ** 0.5))))
    if "hmc" in which: jobs.append(("HMC L=10", 10, dict(sampler=L.SAMPLER_HMC, leapstep=0.1 * (256 / d) ** 0.25, nleaps=10)))
    for name, grads, kw in jobs:
        spl = 32 if grads == 1 else 4
        e = K.Engine(target=K.GaussDenseTarget.compound_symmetric(d, 0.5), nchains=n, nsteps=10 ** 6, steps_per_launch=spl, **kw)
        e.init_state_normal(); e.run(2 * spl)
        r = []
        for _ in range(3):
            k = 4 * spl
            e.run(k); ms, nl = e.last_run_ms(); r.append(n * k / (ms * 1e-3))
        unit = "transitions/s" if grads == 1 else "leapfrog*chain/s"
        print(f"{tag} dense {name} D={d} layout {e.layout()}: {unit} " + " ".join(f"{v * grads:.4g}" for v in r)
              + "  TFLOP/s " + " ".join(f"{v * grads * 2 * d * d / 1e12:.1f}" for v in r)
              + f"  acceptance {e.accept_counts()[0].mean() / max(e.accept_counts()[1], 1):.3f}", flush=True)
        e.close()
