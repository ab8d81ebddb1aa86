#!/usr/bin/env python3
"""Throughput of user-defined targets (run-time compiled closures: whole-vector closures, one chain per lane up to 32 dimensions and staged through LDS on 4-32 lanes beyond; pair closures): the README target written as a closure and the
quartic chain of tests/cases.py, MALA, 65,536 chains."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import cases
import klara_jl_amd as K
from klara_jl_amd import _lib as L

import os
n = 65536
for d, src, data, h in ((16, cases.SRC_NEGDOT, None, 0.1), (32, cases.SRC_NEGDOT, None, 0.1), (40, cases.SRC_NEGDOT, None, 0.1), (100, cases.SRC_NEGDOT, None, 0.05),
                        (100, cases.SRC_QUARTIC_CHAIN, [0.02, 0.5], 0.02), (256, cases.SRC_QUARTIC_CHAIN, [0.02, 0.5], 0.01)):
    for lanes in (("0", "1") if d > 32 else ("0",)):          # beyond 32 dimensions: staged through LDS (default) against one chain per lane (scratch)
        os.environ["KLARA_CUSTOM_LANES"] = lanes
        e = K.Engine(sampler=L.SAMPLER_MALA, target=K.CustomTarget(d, src, data), nchains=n, nsteps=10 ** 6, driftstep=h, monitor=L.MON_SUMMARIES)
        e.init_state_normal(); e.run(32)
        e.run(128); ms, nl = e.last_run_ms()
        print(f"whole-vector closure D = {d:3d} ({'quartic chain' if data else 'negdot'}), layout {e.layout()}: {n * 128 / (ms * 1e-3):.4g} transitions/s "
              f"(kernel registers / scratch / LDS {e.kernel_attributes(0, 32)})", flush=True)
        e.close()
os.environ.pop("KLARA_CUSTOM_LANES", None)

# pair closures (K.CustomTarget.pairwise): the same README closure, a quartic with within-pair coupling and the banana, on the few-lanes-per-chain
# kernels (layout kind 3), next to the built-in diagonal family on the same lanes
print("-- pair closures on the pair-transposed layout, MALA, 65,536 chains, running sums on")
for label, target, h in (("built-in diagonal family, D = 100", K.GaussDiagTarget.negdot(100), 0.9),
                         ("pair closure -dot(z, z), D = 100", K.CustomTarget.pairwise(100, cases.SRC_PAIR_NEGDOT), 0.9),
                         ("pair closure quartic (coupled pairs), D = 100", K.CustomTarget.pairwise(100, cases.SRC_PAIR_QUARTIC, [0.02, 0.5]), 0.3),
                         ("pair closure banana, D = 100", K.CustomTarget.pairwise(100, cases.SRC_PAIR_BANANA, [0.05, 9.0]), 0.2),
                         ("pair closure quartic, D = 300", K.CustomTarget.pairwise(300, cases.SRC_PAIR_QUARTIC, [0.02, 0.5]), 0.1)):
    e = K.Engine(sampler=L.SAMPLER_MALA, target=target, nchains=n, nsteps=10 ** 6, driftstep=h, monitor=L.MON_SUMMARIES, sparse_moves=2)
    e.init_state_normal(); e.run(128)
    e.run(512); ms, nl = e.last_run_ms()
    _, _, na, nt, _ = e.pooled_summaries(with_sums=False)
    print(f"{label}: {n * 512 / (ms * 1e-3):.4g} transitions/s (layout {e.layout()}, acceptance {na / nt:.3f}, kernel {e.kernel_attributes(1, 32)})")
    e.close()
e = K.Engine(sampler=L.SAMPLER_HMC, target=K.CustomTarget.pairwise(100, cases.SRC_PAIR_BANANA, [0.05, 9.0]), nchains=n, nsteps=10 ** 6, leapstep=0.1, nleaps=10)
e.init_state_normal(); e.run(32); e.run(128); ms, nl = e.last_run_ms()
print(f"pair closure banana, HMC L = 10, D = 100: {n * 128 * 10 / (ms * 1e-3):.4g} leapfrog chain/s")
e.close()
