#!/usr/bin/env python3
"""The slice sampler on a pair closure (the coupled quartic of tests/cases.py), D = 100, 65,536 chains: coordinate updates/s on the few-lanes kernels
(k_diagt<SLICE, .., USERPAIR>, round 6) and as a whole-vector closure (KLARA_PAIR_SLICE_AS_WHOLE=1, round 5).   usage: ab_pair_slice.py [tag] [D ...]"""
import os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import klara_jl_amd as K
from klara_jl_amd import _lib as L
import cases

tag = sys.argv[1] if len(sys.argv) > 1 else ""
whole = "KLARA_PAIR_SLICE_AS_WHOLE" in os.environ
for d in [int(v) for v in sys.argv[2:]] or [100]:
    n = 65536 if not whole else 16384
    for name, src, data in (("quartic", cases.SRC_PAIR_QUARTIC, [0.1, 0.4]), ("negdot", cases.SRC_PAIR_NEGDOT, None)):
        e = K.Engine(sampler=L.SAMPLER_SLICE, target=K.CustomTarget.pairwise(d, src, data), nchains=n, nsteps=10 ** 6, slice_widths=np.full(d, 1.0), steps_per_launch=0)
        e.init_state_normal(); e.run(8 if whole else 64)
        r = []
        for _ in range(3):
            k = 8 if whole else 128
            e.run(k); ms, nl = e.last_run_ms(); r.append(n * k / (ms * 1e-3))
        print(f"{tag} slice on pair closure {name} D={d} layout {e.layout()}: chain*transitions/s " + " ".join(f"{v:.4g}" for v in r)
              + "  coordinate updates/s " + " ".join(f"{v * d:.4g}" for v in r), flush=True)
        e.close()
