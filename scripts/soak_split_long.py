#!/usr/bin/env python3
"""Longer race check of the split dense kernels: 65,533 chains, 1,200 transitions of tuned MALA / 300 of HMC with dual averaging / 3 of the slice sampler, twice with different launch cuts: bit-identical ends."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import cases
import klara_jl_amd as K
from klara_jl_amd import _lib as L
n = 65533
for d in (320, 700):
    tgt = K.GaussDenseTarget(cases.compound_symmetric_precision(d, 0.4), const=0.3)
    sc = 256.0 / d
    for name, kw, steps in (("MALA", dict(sampler=L.SAMPLER_MALA, driftstep=0.3 * sc ** (1.0 / 3.0), tuner=L.TUNER_ACCEPT_RATE, targetrate=0.574, period=50), 1200),
                            ("HMC dual averaging", dict(sampler=L.SAMPLER_HMC, leapstep=0.3 * sc ** 0.25, nleaps=3, tuner=L.TUNER_DUAL_AVERAGING, targetrate=0.65, da_nadapt=100), 300),
                            ("slice", dict(sampler=L.SAMPLER_SLICE, slice_widths=np.full(d, 1.5)), 2 if d > 400 else 3)):
        nn = n if name != "slice" else 2045
        outs = []
        for spl, cuts in ((0, [steps]), (7, [1, steps - 1])):
            e = K.Engine(target=tgt, nchains=nn, nsteps=steps, steps_per_launch=spl, monitor=L.MON_SUMMARIES, seed=99 + d, **kw)
            e.init_state_normal()
            for k in cuts:
                e.run(k)
            x, lt, g = e.state(); na, _ = e.accept_counts(); s, q, _ = e.chain_sums()
            e.close()
            outs.append((x, lt, na, s, q))
        same = all(np.array_equal(a, b) for a, b in zip(*outs))
        print(f"D={d} {name}, {steps} transitions, {nn} chains: two launch patterns {'identical' if same else 'DIFFERENT'}; acceptance {outs[0][2].mean() / steps:.3f}", flush=True)
        assert same
