#!/usr/bin/env python3
"""Slice sampler on the README target, D = 100: coordinate updates/s from the library's HIP events.
KLARA_HIP_LIB selects the build, KLARA_SLICE_LOCKSTEP=1 round 4's lockstep kernel, KLARA_SLICE_MACHINES=1|2 the free-running kernel's machines per lane;
AB_SPL (transitions per launch, default 32), AB_CHAINS (default 65536), AB_WIDTH (slice width, default 1.0)."""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import klara_jl_amd as K
from klara_jl_amd import _lib as L
n, d = int(os.environ.get("AB_CHAINS", "65536")), 100
spl, width = int(os.environ.get("AB_SPL", "32")), float(os.environ.get("AB_WIDTH", "1.0"))
tag = f"[{os.path.basename(os.environ.get('KLARA_HIP_LIB', 'default'))} lockstep={os.environ.get('KLARA_SLICE_LOCKSTEP', '0')} machines={os.environ.get('KLARA_SLICE_MACHINES', '-')} spl={spl} chains={n} width={width}]"
for mon in (0, L.MON_SUMMARIES):
    e = K.Engine(sampler=L.SAMPLER_SLICE, target=K.GaussDiagTarget.negdot(d), nchains=n, nsteps=10 ** 7, slice_widths=np.full(d, width), steps_per_launch=spl, monitor=mon, nstreams=1)
    e.init_state_normal(); e.run(2 * spl)
    r = []
    for _ in range(3):
        e.run(4 * spl); ms, nl = e.last_run_ms(); r.append(n * 4 * spl * d / (ms * 1e-3))
    print(f"{tag} slice D={d} {'running sums' if mon else 'no monitor'}: coordinate updates/s " + " ".join(f"{v:.4g}" for v in r), flush=True)
    e.close()
