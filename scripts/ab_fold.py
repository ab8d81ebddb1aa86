#!/usr/bin/env python3
"""Same-box A/B of library builds on the 4-lane MALA kernel with running sums (sums folded in memory) against the acceptance rate: steady-state
us per transition, one stream, median of 5 runs of 640 transitions.  KLARA_HIP_LIB selects the build, AB_TAG labels the line."""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import klara_jl_amd as K
from klara_jl_amd import _lib as L

tag = os.environ.get("AB_TAG", os.path.basename(os.environ.get("KLARA_HIP_LIB", "default")))
out = []
for h in (2.0, 0.9, 0.75, 0.7, 0.6):
    e = K.Engine(sampler=L.SAMPLER_MALA, target=K.GaussDiagTarget.negdot(100), nchains=65536, nsteps=10 ** 7, driftstep=h, monitor=L.MON_SUMMARIES, sparse_moves=1, nstreams=1)
    e.init_state_normal(); e.run(640)
    ts = []
    for _ in range(5):
        e.run(640); ts.append(e.last_run_ms()[0] * 1e3 / 640)
    out.append(f"h={h}: {np.median(ts):.2f}")
    e.close()
print(f"[{tag}] " + "; ".join(out), flush=True)
