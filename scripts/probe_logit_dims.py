#!/usr/bin/env python3
"""Logistic regression beyond 16 parameters: MALA / HMC (L = 10) / MH transitions/s on 200 (and 1,000) synthetic rows, 32,768 chains with running sums, from the
library's HIP events; FP64 TFLOP/s of the two MFMA passes (4 n D flop per gradient evaluation and chain, 2 n D for MH's log-target alone).  KLARA_LOGIT_NO_MFMA=1 in
the environment gives the closure form of rounds 1-5 (one chain per lane)."""
import os, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import cases
import klara_jl_amd as K
from klara_jl_amd import _lib as L

tag = "closure" if "KLARA_LOGIT_NO_MFMA" in os.environ else os.environ.get("AB_TAG", "mfma")
n = int(os.environ.get("AB_CHAINS", "32768"))
for d, nd in [tuple(int(v) for v in t.split('x')) for t in os.environ.get('AB_SHAPES', '16x200,20x200,32x200,64x200,128x200,64x1000').split(',')]:
    X, y = cases.synthetic_logit(nd, d)
    for name, kw, evals, fl in (("MALA", dict(sampler=L.SAMPLER_MALA, driftstep=0.02), 1, 4), ("HMC L=10", dict(sampler=L.SAMPLER_HMC, leapstep=0.02, nleaps=10), 10, 4),
                                ("MH", dict(sampler=L.SAMPLER_MH, mh_sigma=np.full(d, 0.02)), 1, 2), ("SLICE", dict(sampler=L.SAMPLER_SLICE, slice_widths=np.full(d, 1.0)), 0, 0)):
        if (tag == "closure" and name not in ("MALA", "SLICE")) or name.split()[0] not in os.environ.get("AB_SAMPLERS", "MALA,HMC,MH,SLICE").split(","):
            continue
        if name == "SLICE" and d == 16:
            continue
        spl, nrun = (1, 2) if name == "SLICE" else (8, 16)          # (slice: D coordinate updates of a few full evaluations each per transition)
        e = K.Engine(target=K.LogisticTarget(X, y, 10.0), nchains=n, nsteps=10 ** 6, monitor=L.MON_SUMMARIES, steps_per_launch=spl, **kw)
        e.set_state(0.1 * np.random.default_rng(0).standard_normal((n, d))); e.run(spl)
        r = []
        for _ in range(2):
            e.run(nrun); ms, nl = e.last_run_ms(); r.append(n * nrun / (ms * 1e-3))
        tf = np.median(r) * evals * fl * nd * d / 1e12
        tail = f"  coordinate updates/s {np.median(r) * d:.4g}" if name == "SLICE" else f"  {tf:.1f} TFLOP/s"
        print(f"{tag} logistic {name} D={d} n={nd} layout {e.layout()}: transitions/s " + " ".join(f"{v:.4g}" for v in r) + tail, flush=True)
        e.close()
