#!/usr/bin/env python3
"""Same-box A/B of library builds (KLARA_HIP_LIB selects the .so) on cfg 4 (swiss logistic MALA, 32,768 chains, running sums, 50 transitions
per launch) and cfg 5 (rats HMC L = 32, 131,072 chains, pooled tuner) exactly as bench.py's extras set them up: transitions/s, median of 5."""
import os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import klara_jl_amd as K
from klara_jl_amd import _lib as L

tag = os.environ.get("AB_TAG", os.path.basename(os.environ.get("KLARA_HIP_LIB", "default")))
which = os.environ.get("AB_CFG", "4,5").split(",")
gold = ROOT / "tests" / "golden"


def rate_of(e, n, steps, warm):
    e.run(warm)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); e.run(steps); ts.append(time.perf_counter() - t0)
    return n * steps / np.median(ts)


if "4" in which:
    sw = np.load(gold / "swiss.npz")
    X = sw["measurements"]; X = np.ascontiguousarray((X - X.mean(axis=0)) / X.std(axis=0, ddof=1))
    y = np.ascontiguousarray(sw["status"].astype(np.float64))
    nc = 32768
    x0 = np.array([5.1, -0.9, 8.2, -4.5])[None, :] + 0.1 * np.random.default_rng(0).standard_normal((nc, 4))
    for samp, kw in ((L.SAMPLER_MALA, dict(driftstep=0.1)), (L.SAMPLER_MH, dict(mh_sigma=[0.1] * 4)), (L.SAMPLER_HMC, dict(leapstep=0.05, nleaps=4))):
        e = K.Engine(sampler=samp, target=K.LogisticTarget(X, y, 100.0), nchains=nc, nsteps=10 ** 6, burnin=1000, steps_per_launch=50,
                     monitor=L.MON_SUMMARIES, **kw)
        e.set_state(x0)
        r = rate_of(e, nc, 100, 500)
        print(f"[{tag}] cfg4 logistic, sampler {samp}: {r:.4g} transitions/s", flush=True)
        e.close()
if "5" in which:
    rats = np.load(gold / "rats.npz")
    t = K.HierNormalTarget(rats["weight"], rats["age"] - 22.0)
    nc = 131072
    x0 = t.least_squares_start()[None, :] + 0.05 * np.random.default_rng(1).standard_normal((nc, t.ndims))
    e = K.Engine(sampler=L.SAMPLER_HMC, target=t, nchains=nc, nsteps=2000, burnin=1000, leapstep=0.02, nleaps=32,
                 tuner=L.TUNER_ACCEPT_RATE, tuner_mode=L.TUNE_POOLED, targetrate=0.65, period=100, steps_per_launch=10, monitor=L.MON_SUMMARIES)
    e.set_state(x0)
    r = rate_of(e, nc, 100, 200)
    print(f"[{tag}] cfg5 rats HMC L=32: {r * 32:.4g} leapfrog chain/s", flush=True)
    e.close()
if "slice" in which:
    for D, w in ((100, 1.0), (40, 1.0), (200, 1.0)):
        n = 65536
        e = K.Engine(sampler=L.SAMPLER_SLICE, target=K.GaussDiagTarget.negdot(D), nchains=n, nsteps=10 ** 7, slice_widths=np.full(D, w), steps_per_launch=4, nstreams=1)
        e.init_state_normal()
        r = rate_of(e, n, 16, 4)
        print(f"[{tag}] slice sampler, lt = -|x|^2, D = {D}: {r:.4g} transitions/s = {r * D:.4g} coordinate updates/s", flush=True)
        e.close()
    e = K.Engine(sampler=L.SAMPLER_SLICE, target=K.GaussDiagTarget(100, w=np.linspace(0.5, 2.0, 100), mu=np.linspace(-1.0, 1.0, 100)), nchains=65536, nsteps=10 ** 7,
                 slice_widths=np.full(100, 1.0), steps_per_launch=4, nstreams=1, monitor=L.MON_SUMMARIES)
    e.init_state_normal()
    r = rate_of(e, 65536, 16, 4)
    print(f"[{tag}] slice sampler, weighted diagonal with means, running sums, D = 100: {r:.4g} transitions/s", flush=True)
    e.close()
