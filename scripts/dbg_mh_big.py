#!/usr/bin/env python3
"""Debug: MH on the streamed dense layout, NE = 48 with a mean vector."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import klara_jl_amd as K
from klara_jl_amd import _lib as L
import oracle_ffi as O
import cases
import test_gpu_parity as T

for seed in (10, 11):
    c, rng = T._random_case(seed, wide=True)
    for spl in (0, 1, 5):
        for ns in (0, 1, 3):
            mon = L.MON_ACCEPT | L.MON_SUMMARIES
            eng = K.Engine(**cases.engine_kwargs(c, monitor=mon, steps_per_launch=spl, nstreams=ns))
            job = O.OracleJob(**cases.oracle_kwargs(c, layout=eng.layout()))
            if c["x0"] is None:
                eng.init_state_normal(); job.init_state_normal()
            else:
                eng.set_state(c["x0"]); job.set_state(c["x0"])
            x, lt, g = eng.state()
            same0 = np.array_equal(x, job.X) and np.array_equal(lt, job.LT)
            eng.run(c["nsteps"]); job.run(c["nsteps"])
            m = eng.accept_mask()
            bad = np.argwhere(m != job.accept)
            x, lt, g = eng.state()
            print(seed, "spl", spl, "ns", ns, "layout", eng.layout(), "init same", same0, "mismatches", len(bad), bad[:3].tolist(), "x same", np.array_equal(x, job.X), flush=True)
            eng.close()
