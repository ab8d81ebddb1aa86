#!/usr/bin/env python3
"""Per-transition cost of the headline kernels against the acceptance rate (MALA on lt = -|x|^2, D = 100, 65,536 chains, library-default
fusion, steady state after 600 transitions): the drift step sets the acceptance; without the save rule, with it (8-lane kernels, resident sums) and with it under the
`sparse_moves` hint (4-lane kernels, atomic folds)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import klara_jl_amd as K
from klara_jl_amd import _lib as L

for mon, sparse in ((0, 0), (L.MON_SUMMARIES, 0), (L.MON_SUMMARIES, 1)):
    for h in (0.9, 0.5, 0.3, 0.1, 0.01):
        for spl in (32, 1):
            e = K.Engine(sampler=L.SAMPLER_MALA, target=K.GaussDiagTarget.negdot(100), nchains=65536, nsteps=10 ** 7, driftstep=h,
                         monitor=mon, sparse_moves=sparse, steps_per_launch=spl, nstreams=1)
            e.init_state_normal(); e.run(640)
            _, _, na0, nt0, _ = e.pooled_summaries(with_sums=False)
            e.run(640)
            ms, nl = e.last_run_ms()
            _, _, na, nt, _ = e.pooled_summaries(with_sums=False)
            mode = "off" if not mon else ("on, sparse_moves hint (4 lanes, atomic folds)" if sparse else "on (8 lanes, resident sums)")
            print(f"save={mode} drift {h:5.2f} spl {spl:2d}: {ms * 1e3 / 640:6.2f} us per transition, acceptance {(na - na0) / (nt - nt0):.4f}")
            e.close()
