#!/usr/bin/env python3
"""The driver-shaped region (fresh bench job, run(5), then 5 x run(20) timed by the host clock) for several ways of cutting a 20-transition run into
launches: steps_per_launch 0 (library default: one launch on the caller's stream), 20, 10, 7, 5, 4 (multi-launch: two chain partitions on two streams).
A scratch job keeps the device busy before each repetition, as bench.py does."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import klara_jl_amd as K
from klara_jl_amd import _lib as L

neg = K.GaussDiagTarget.negdot(100)
w = K.Engine(sampler=L.SAMPLER_MALA, target=neg, nchains=65536, nsteps=10 ** 7, driftstep=0.9, monitor=0); w.init_state_normal(); w.run(3200)
for spl in [int(v) for v in (sys.argv[1:] or ["0", "20", "10", "7", "5", "4", "0"])]:
    for rep in range(2):
        e = K.Engine(sampler=L.SAMPLER_MALA, target=neg, nchains=65536, nsteps=10 ** 7, driftstep=0.9, seed=20260927, monitor=L.MON_SUMMARIES, steps_per_launch=spl)
        e.init_state_normal(); e.run(5)
        w.run(3200)
        ts, ks = [], []
        for r in range(5):
            t0 = time.perf_counter(); e.run(20); ts.append((time.perf_counter() - t0) / 20 * 1e6)
            ms, nl = e.last_run_ms(); ks.append(ms * 1e3 / 20)
            w.run(320)
        print(f"steps_per_launch {spl:2d}: wall us/transition {' '.join('%.2f' % t for t in ts)} median {np.median(ts):.2f} | kernel {' '.join('%.2f' % k for k in ks)} | launch kinds {tuple(int(v) for v in e.launch_modes()[0])}", flush=True)
        e.close()
w.close()
