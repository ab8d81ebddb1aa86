#!/usr/bin/env python3
"""HMC with dual averaging on the split dense layout: a few launches for a counter pass (executed matrix instructions per second against the plain kernel's)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import klara_jl_amd as K
from klara_jl_amd import _lib as L
d = int(sys.argv[1]) if len(sys.argv) > 1 else 512
for tag, kw in (("plain", {}), ("da", dict(tuner=L.TUNER_DUAL_AVERAGING, targetrate=0.7, da_nadapt=10 ** 6))):
    e = K.Engine(target=K.GaussDenseTarget.compound_symmetric(d, 0.5), nchains=65536, nsteps=10 ** 6, steps_per_launch=4, sampler=L.SAMPLER_HMC,
                 leapstep=0.1 * (256 / d) ** 0.25, nleaps=10, **kw)
    e.init_state_normal(); e.run(8)
    for _ in range(3):
        e.run(16); ms, nl = e.last_run_ms(); print(tag, d, "ms per 16 transitions", ms, flush=True)
    e.close()
