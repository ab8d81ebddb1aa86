#!/usr/bin/env python3
"""Soak / race check: the same job run twice (two chain partitions on two streams, one transition per launch, 20,000
launches each) must end in bit-identical states; also against one stream and against fused launches."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import klara_jl_amd as K
from klara_jl_amd import _lib as L

def run(nstreams, spl, steps, sampler=L.SAMPLER_MALA, **kw):
    e = K.Engine(sampler=sampler, target=K.GaussDiagTarget.negdot(100), nchains=65536 - 3, nsteps=10 ** 7, steps_per_launch=spl, monitor=0,
                 nstreams=nstreams, **kw)
    e.init_state_normal()
    for _ in range(4):
        e.run(steps // 4)
    x, lt, g = e.state(); na, _ = e.accept_counts()
    e.close()
    return x, lt, g, na

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
ref = run(0, 1, steps, driftstep=0.3)
for label, args in (("two streams again", (0, 1)), ("one stream", (1, 1)), ("fused 16/launch", (0, 16)), ("three streams", (3, 1))):
    out = run(args[0], args[1], steps, driftstep=0.3)
    ok = all(np.array_equal(a, b) for a, b in zip(ref, out))
    print(f"MALA {steps} transitions, {label:18s}: {'identical' if ok else 'DIFFERENT'}; acceptance {out[3].mean() / steps:.3f}")
    assert ok
h1 = run(0, 1, steps // 10, sampler=L.SAMPLER_HMC, leapstep=0.1, nleaps=10)
h2 = run(0, 7, steps // 10, sampler=L.SAMPLER_HMC, leapstep=0.1, nleaps=10)
assert all(np.array_equal(a, b) for a, b in zip(h1, h2)); print("HMC: identical across launch patterns")
x = ref[0]; print("ensemble mean/var of final x:", float(x.mean()), float(x.var()))
