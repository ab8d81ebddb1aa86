#!/usr/bin/env python3
"""BASELINE configs[0] batched: MH sigma=(1,1) on lt = -|x|^2, D = 2, N replicas of the README chain (test/README.jl:5-47)."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import klara_jl_amd as K
from klara_jl_amd import _lib as L

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
for spl in (1, 16, 100):
    e = K.Engine(sampler=L.SAMPLER_MH, target=K.GaussDiagTarget.negdot(2), nchains=n, nsteps=10 ** 6, burnin=1000,
                 mh_sigma=[1.0, 1.0], steps_per_launch=spl, monitor=0)
    e.set_state(np.tile(np.array([5.1, -0.9]), (n, 1))); e.run(200)
    t0 = time.perf_counter(); e.run(2000); dt = time.perf_counter() - t0
    print(f"cfg1 MH D=2, {n} chains, {spl} per launch, layout {e.layout()}: {n * 2000 / dt:.4g} transitions/s")
    e.close()
e = K.Engine(sampler=L.SAMPLER_MH, target=K.GaussDiagTarget.negdot(2), nchains=n, nsteps=10000, burnin=1000,
             mh_sigma=[1.0, 1.0], monitor=L.MON_SUMMARIES)
e.set_state(np.tile(np.array([5.1, -0.9]), (n, 1)))
t0 = time.perf_counter(); e.run(10000); dt = time.perf_counter() - t0
s, q, ns = e.chain_sums()
print(f"README job x {n}: {dt:.3f} s, mean of chain means {s.mean(0) / ns}, var {(q.sum(0) / (ns * n))}")
