#!/usr/bin/env python3
"""One line per configuration for same-box A/B runs of two library builds (KLARA_HIP_LIB): cfg 3 dense HMC and cfg 5 rats HMC,
kernel time from the library's own HIP events.   usage: KLARA_HIP_LIB=... python scripts/ab_models.py [tag]"""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import cases
import klara_jl_amd as K
from klara_jl_amd import _lib as L

tag = sys.argv[1] if len(sys.argv) > 1 else ""
n = 65536
e = K.Engine(sampler=L.SAMPLER_HMC, target=K.GaussDenseTarget.compound_symmetric(100), nchains=n, nsteps=10 ** 6, leapstep=0.1, nleaps=10,
             monitor=L.MON_SUMMARIES)
e.init_state_normal(); e.run(64)
r = []
for _ in range(3):
    e.run(128); ms, nl = e.last_run_ms(); r.append(n * 128 * 10 / (ms * 1e-3))
print(f"{tag} cfg3 dense HMC L=10 : leapfrog*chain/s " + " ".join(f"{v:.4g}" for v in r))
e.close()

t = cases.rats_target()
n = 131072
x0 = t.least_squares_start()[None, :] + 0.05 * np.random.default_rng(1).standard_normal((n, t.ndims))
e = K.Engine(sampler=L.SAMPLER_HMC, target=t, nchains=n, nsteps=2000, burnin=1000, leapstep=0.02, nleaps=32,
             tuner=L.TUNER_ACCEPT_RATE, tuner_mode=L.TUNE_POOLED, targetrate=0.65, period=100, monitor=L.MON_SUMMARIES)
e.set_state(x0); e.run(100)
r = []
for _ in range(3):
    e.run(100); ms, nl = e.last_run_ms(); r.append(n * 100 * 32 / (ms * 1e-3))
print(f"{tag} cfg5 rats HMC L=32  : leapfrog*chain/s " + " ".join(f"{v:.4g}" for v in r))
e.close()
