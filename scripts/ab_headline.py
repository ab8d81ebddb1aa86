#!/usr/bin/env python3
"""Same-box A/B of library builds on the headline kernel alone: steady-state us per transition of the 4-lane MALA kernel (one stream), without and
with running sums, at two acceptance rates; median of 5 runs of 640 transitions.  KLARA_HIP_LIB selects the build, AB_TAG labels the line."""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import klara_jl_amd as K
from klara_jl_amd import _lib as L

tag = os.environ.get("AB_TAG", os.path.basename(os.environ.get("KLARA_HIP_LIB", "default")))
N, D = 65536, 100
out = []
for mon, mode in ((0, 0), (L.MON_SUMMARIES, 1), (L.MON_SUMMARIES, 2)):
    for h in (0.9, 0.6):
        e = K.Engine(sampler=L.SAMPLER_MALA, target=K.GaussDiagTarget.negdot(D), nchains=N, nsteps=10 ** 7, driftstep=h, monitor=mon, sparse_moves=mode, nstreams=1)
        e.init_state_normal(); e.run(640)
        ts = []
        for _ in range(5):
            e.run(640); ts.append(e.last_run_ms()[0] * 1e3 / 640)
        out.append(f"{'no save' if not mon else ('4-lane' if mode == 1 else '8-lane')} h={h}: {np.median(ts):.2f}")
        e.close()
print(f"[{tag}] " + "; ".join(out), flush=True)
