#!/usr/bin/env python3
"""One bench-shaped job for a kernel trace: run(5), scratch load, then 6 x run(20) with scratch load between (rocprofv3 --kernel-trace shows what a 20-transition run launches)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import klara_jl_amd as K
from klara_jl_amd import _lib as L
neg = K.GaussDiagTarget.negdot(100)
w = K.Engine(sampler=L.SAMPLER_MALA, target=neg, nchains=65536, nsteps=10 ** 7, driftstep=0.9, monitor=0); w.init_state_normal(); w.run(640)
e = K.Engine(sampler=L.SAMPLER_MALA, target=neg, nchains=65536, nsteps=10 ** 7, driftstep=0.9, seed=20260927, monitor=L.MON_SUMMARIES)
e.init_state_normal(); e.run(5); w.run(640)
for r in range(6):
    t0 = time.perf_counter(); e.run(20); dt = time.perf_counter() - t0
    ms, nl = e.last_run_ms()
    print(f"run(20) #{r}: wall {dt * 1e6:.1f} us, events {ms * 1e3:.1f} us, launches {nl}")
    w.run(320)
for n in (1, 2, 4, 8, 12, 16, 20, 24, 28, 32):
    ts = []
    for r in range(5):
        e.run(n); ms, nl = e.last_run_ms(); ts.append(ms * 1e3); w.run(64)
    print(f"steady run({n}): events us " + " ".join(f"{t:.1f}" for t in ts))
