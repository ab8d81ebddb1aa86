import sys, time
from pathlib import Path
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import cases
import klara_jl_amd as K
from klara_jl_amd import _lib as L
n = 65536
for d in (100, 130):
    t0 = time.time()
    e = K.Engine(sampler=L.SAMPLER_MALA, target=K.CustomTarget(d, cases.SRC_NEGDOT, None), nchains=n, nsteps=10 ** 6, driftstep=0.05)
    tc = time.time() - t0
    e.init_state_normal(); e.run(32)
    e.run(128); ms, nl = e.last_run_ms()
    print(f"custom closure D = {d}: create {tc:.1f} s, {n * 128 / (ms * 1e-3):.4g} transitions/s")
    e.close()
