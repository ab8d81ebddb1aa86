#!/usr/bin/env python3
"""HMC L=10 eps=0.1 on the README target (-|x|^2, D=100), 65,536 chains: leapfrog*chain/s at 1 and 16 transitions per launch."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import klara_jl_amd as K
from klara_jl_amd import _lib as L

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
for spl, steps in ((1, 400), (16, 640)):
    e = K.Engine(sampler=L.SAMPLER_HMC, target=K.GaussDiagTarget.negdot(100), nchains=n, nsteps=10 ** 6, leapstep=0.1, nleaps=10,
                 steps_per_launch=spl, monitor=0)
    e.init_state_normal(); e.run(2 * spl)
    t0 = time.perf_counter(); e.run(steps); dt = time.perf_counter() - t0
    ms, nl = e.last_run_ms()
    print(f"HMC iso layout {e.layout()} spl={spl}: {n * steps * 10 / dt:.4g} leapfrog*chain/s, {ms / nl * 1e3 / spl:.1f} us per transition of all chains")
    e.close()
