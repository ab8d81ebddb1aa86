"""Host cost of enqueueing 2,000 one-transition launches (klara_run_async) against the time the GPU needs to run them."""
import sys, time
from pathlib import Path; sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import klara_jl_amd as K
from klara_jl_amd import _lib as L
e = K.Engine(sampler=L.SAMPLER_MALA, target=K.GaussDiagTarget.negdot(100), nchains=65536, nsteps=10**6, driftstep=0.9, steps_per_launch=1, monitor=0)
e.init_state_normal(); e.run(100)
lib = L.load()
t0 = time.perf_counter(); L.check(lib.klara_run_async(e._h, 2000), "run_async"); t1 = time.perf_counter(); L.check(lib.klara_synchronize(e._h), "sync"); t2 = time.perf_counter()
print(f"enqueue 2000 steps (4000 launches): {(t1-t0)*1e3:.1f} ms = {(t1-t0)/2000*1e6:.1f} us per step; total {(t2-t0)*1e3:.1f} ms")
