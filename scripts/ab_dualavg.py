#!/usr/bin/env python3
"""HMC with the dual-averaging tuner on several layouts (dense D = 100 / 128 / 256, logistic regression on the matrix cores, the rats model, the diagonal Gaussian):
ms per 16 transitions in the first three runs of a fresh job (the step adapts: the runs differ, the builds see the same sequence).  KLARA_HIP_LIB selects the build."""
import os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import klara_jl_amd as K
from klara_jl_amd import _lib as L
import cases
tag = sys.argv[1] if len(sys.argv) > 1 else os.path.basename(os.environ.get("KLARA_HIP_LIB", "default"))
da = dict(tuner=L.TUNER_DUAL_AVERAGING, targetrate=0.7, da_nadapt=10 ** 6)
jobs = []
for d in (100, 128, 256):
    jobs.append((f"dense D={d}", 65536, dict(target=K.GaussDenseTarget.compound_symmetric(d, 0.5), leapstep=0.1 * (100 / d) ** 0.25, nleaps=10)))
X, y = cases.synthetic_logit(200, 64, seed=3)
jobs.append(("logistic D=64 n=200", 32768, dict(target=K.LogisticTarget(X / 8.0, y, 10.0), leapstep=0.05, nleaps=10, x0scale=0.1)))
jobs.append(("diag D=100", 65536, dict(target=K.GaussDiagTarget.negdot(100), leapstep=0.1, nleaps=10)))
rats = cases.make_case("hmc_rats_dualavg")
jobs.append(("rats", 32768, dict(target=rats["target"], leapstep=rats["leapstep"], nleaps=rats["nleaps"], rats=True)))
for name, n, kw in jobs:
    x0s = kw.pop("x0scale", None); is_rats = kw.pop("rats", False)
    e = K.Engine(sampler=L.SAMPLER_HMC, nchains=n, nsteps=10 ** 6, steps_per_launch=4, **kw, **da)
    if is_rats:
        e.set_state(kw["target"].least_squares_start()[None, :] + 0.02 * np.random.default_rng(1).standard_normal((n, kw["target"].ndims)))
    elif x0s:
        e.set_state(x0s * np.random.default_rng(1).standard_normal((n, kw["target"].ndims)))
    else:
        e.init_state_normal()
    e.run(8)
    r = []
    for _ in range(3):
        e.run(16); ms, nl = e.last_run_ms(); r.append(ms)
    print(f"[{tag}] HMC dual averaging, {name}, layout {e.layout()}: ms per 16 transitions " + " ".join(f"{v:.2f}" for v in r) + f"  regs/scratch {e.kernel_attributes(0, 4)[:2]}", flush=True)
    e.close()
