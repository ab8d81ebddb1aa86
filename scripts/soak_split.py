#!/usr/bin/env python3
"""Race check of the workgroup-split dense kernels (klara_dense_split.h: every transition crosses several workgroup barriers and an LDS exchange) at
full occupancy: the same job — 65,533 chains — run twice with different launch cuts must end in bit-identical states, and blocks of chains at the
start, in the middle and at the ragged end are replayed by the oracle."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import cases, oracle_ffi as O
import klara_jl_amd as K
from klara_jl_amd import _lib as L

n = 65533
for d, mean in ((320, False), (512, True), (1024, False)):
    rng = np.random.default_rng(d)
    tgt = K.GaussDenseTarget(cases.compound_symmetric_precision(d, 0.4), const=0.3, mu=(rng.uniform(-1.5, 1.5, d) if mean else None))
    sc = 256.0 / d
    for name, kw, steps in (("MALA", dict(sampler=L.SAMPLER_MALA, driftstep=0.3 * sc ** (1.0 / 3.0), tuner=L.TUNER_ACCEPT_RATE, targetrate=0.574, period=5), 24),
                            ("HMC dual averaging", dict(sampler=L.SAMPLER_HMC, leapstep=0.3 * sc ** 0.25, nleaps=3, tuner=L.TUNER_DUAL_AVERAGING, targetrate=0.65, da_nadapt=12), 16),
                            ("MH", dict(sampler=L.SAMPLER_MH, mh_sigma=np.full(d, 0.07 * sc ** 0.5)), 24)):
        outs = []
        for spl, cuts in ((0, [steps]), (5, [3, steps - 3])):
            e = K.Engine(target=tgt, nchains=n, nsteps=steps, steps_per_launch=spl, monitor=L.MON_SUMMARIES, seed=4242 + d, **kw)
            e.init_state_normal()
            for k in cuts:
                e.run(k)
            x, lt, g = e.state(); na, _ = e.accept_counts(); s, q, _ = e.chain_sums()
            lay = e.layout(); e.close()
            outs.append((x, lt, g, na, s, q))
        same = all(np.array_equal(a, b) for a, b in zip(*outs))
        case = dict(kw, target=tgt, nchains=16, nsteps=steps, name="soak", x0=None, seed=4242 + d, burnin=0, thinning=1)
        ok = True
        if kw.get("tuner_mode", 0) == 0:
            for off in (0, 32768 + 16, n - 13):
                cnt = min(16, n - off)
                job = O.OracleJob(**cases.oracle_kwargs(dict(case, nchains=cnt), layout=lay, chain_offset=off))
                job.init_state_normal(); job.run(steps)
                sl = slice(off, off + cnt)
                ok = ok and np.array_equal(outs[0][0][sl], job.X) and np.array_equal(outs[0][1][sl], job.LT) and np.array_equal(outs[0][3][sl], job.naccept)
        print(f"D={d} {name}: layout {lay}, two launch patterns {'identical' if same else 'DIFFERENT'}, oracle blocks {'identical' if ok else 'DIFFERENT'}, acceptance {outs[0][3].mean() / steps:.3f}", flush=True)
        assert same and ok
