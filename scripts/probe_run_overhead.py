#!/usr/bin/env python3
"""Host-side cost of one klara_run call on the headline job (MALA, lt = -|x|^2, D = 100, 65,536 chains, running sums): wall-clock per
call for runs of 1, 20 and 64 transitions (median over many calls of a job in steady state), next to the kernel time HIP events report."""
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import klara_jl_amd as K
from klara_jl_amd import _lib as L

tag = os.environ.get("AB_TAG", "default")
e = K.Engine(sampler=L.SAMPLER_MALA, target=K.GaussDiagTarget.negdot(100), nchains=65536, nsteps=10 ** 8, driftstep=0.9, seed=20260927, monitor=L.MON_SUMMARIES)
e.init_state_normal(); e.run(3200)
for n in (1, 20, 64):
    ts, ks = [], []
    for r in range(60):
        t0 = time.perf_counter(); e.run(n); ts.append((time.perf_counter() - t0) * 1e6)
        try:
            ks.append(e.last_run_ms()[0] * 1e3)
        except Exception:
            ks.append(float("nan"))
    print(f"[{tag}] run({n}): wall {np.median(ts[10:]):8.2f} us per call = {np.median(ts[10:]) / n:6.2f} us/transition; between the events {np.median(ks[10:]):8.2f} us; "
          f"host + synchronisation {np.median(ts[10:]) - np.median(ks[10:]):6.2f} us", flush=True)
e.close()
