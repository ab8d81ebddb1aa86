#!/usr/bin/env python3
"""Per-transition cost of the headline kernels against the acceptance rate (MALA on lt = -|x|^2, D = 100, 65,536 chains, steady state
after 640 transitions): the drift step sets the acceptance.  Without the save rule (4-lane kernels), and with it in the three ways
klara_desc.sparse_moves selects: 0 = the library decides on the device launch by launch (what a caller gets), 1 = always the 4-lane
kernels (a moving chain's sums folded into memory by a read-modify-write), 2 = always the 8-lane kernels (resident sums).
usage: probe_acceptance_cost.py [--quick]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import klara_jl_amd as K
from klara_jl_amd import _lib as L

quick = "--quick" in sys.argv
import os
drifts = tuple(float(v) for v in os.environ.get("PROBE_DRIFTS", "0.9,0.8,0.7,0.65,0.6,0.55,0.5,0.3,0.1,0.01").split(","))
modes = os.environ.get("PROBE_MODES", "00,10,11,12").split(",")
NAMES = {(0, 0): "off", (1, 0): "on, library decides (sparse_moves = 0)", (1, 1): "on, 4 lanes, sums folded in memory (sparse_moves = 1)",
         (1, 2): "on, 8 lanes + resident sums (sparse_moves = 2)"}
for mon, sparse in [(L.MON_SUMMARIES if m[0] == "1" else 0, int(m[1])) for m in modes]:
    for h in drifts:
        for spl in ((32,) if quick else (32, 1)):
            e = K.Engine(sampler=L.SAMPLER_MALA, target=K.GaussDiagTarget.negdot(100), nchains=65536, nsteps=10 ** 7, driftstep=h,
                         monitor=mon, sparse_moves=sparse, steps_per_launch=spl, nstreams=1)
            e.init_state_normal(); e.run(640)
            _, _, na0, nt0, _ = e.pooled_summaries(with_sums=False)
            c0 = e.launch_modes()[0].copy()
            e.run(640)
            ms, nl = e.last_run_ms()
            _, _, na, nt, _ = e.pooled_summaries(with_sums=False)
            cnt, lm, la = e.launch_modes()
            print(f"save={NAMES[(1 if mon else 0, sparse)]} drift {h:5.2f} spl {spl:2d}: {ms * 1e3 / 640:6.2f} us per transition, "
                  f"acceptance {(na - na0) / (nt - nt0):.4f}, launches (4-lane, 8-lane, device-decided) {tuple(int(v) for v in cnt - c0)}, last decision {int(lm[0])}", flush=True)
            e.close()
