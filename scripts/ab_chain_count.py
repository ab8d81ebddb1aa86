#!/usr/bin/env python3
"""Occupancy quantisation of the headline kernel: us per transition and transitions/s per chain count (4 lanes per chain, 16 chains per wavefront,
3 wavefronts per SIMD resident at 168 registers: 49,152 chains fill the chip exactly once, 65,536 need a second, one-wavefront round)."""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import klara_jl_amd as K
from klara_jl_amd import _lib as L

D = 100
for N in [int(v) for v in os.environ.get("AB_CHAINS", "16384,32768,49152,57344,65536,81920,98304,131072").split(",")]:
    for mon in (0, L.MON_SUMMARIES):
        e = K.Engine(sampler=L.SAMPLER_MALA, target=K.GaussDiagTarget.negdot(D), nchains=N, nsteps=10 ** 7, driftstep=0.9, monitor=mon, nstreams=1)
        e.init_state_normal(); e.run(640)
        ts = []
        for _ in range(5):
            e.run(640); ts.append(e.last_run_ms()[0] * 1e3 / 640)
        t = float(np.median(ts))
        print(f"N={N:7d} {'sums' if mon else 'none'}: {t:7.2f} us/transition  {N / t * 1e6:.3e} tr/s  {t / N * 65536:.2f} us per 65,536", flush=True)
        e.close()
