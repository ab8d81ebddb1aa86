#!/usr/bin/env python3
"""Why is a 20-transition timed region slower per transition than a 1024-transition one?  Separates the two candidates:
the transient of the workload (acceptance 7 % in the first hundred transitions from x0 ~ N(0, I), 0.4 % at stationarity) and the
clock ramp of a GPU that was idle.  A: fresh job, cold GPU.  B: same job after 3,000 more transitions (hot, low acceptance).
C: after 2 s of idling (cold, low acceptance).  D: fresh job straight after sustained load on another one (hot, high acceptance)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import klara_jl_amd as K
from klara_jl_amd import _lib as L


def job():
    e = K.Engine(sampler=L.SAMPLER_MALA, target=K.GaussDiagTarget.negdot(100), nchains=65536, nsteps=10 ** 7, driftstep=0.9,
                 monitor=L.MON_SUMMARIES, sparse_moves=1)                 # the bench job
    e.init_state_normal()
    return e


def reps(e, tag, n=9, steps=20):
    out = []
    for _ in range(n):
        t0 = time.perf_counter(); e.run(steps); dt = time.perf_counter() - t0
        ms, nl = e.last_run_ms()
        out.append((dt * 1e6 / steps, ms * 1e3 / steps))
    _, _, na, nt, _ = e.pooled_summaries(with_sums=False)
    print(tag, "wall us/step", " ".join(f"{a:.1f}" for a, _ in out), "| kernel us/step", " ".join(f"{b:.1f}" for _, b in out), "| acceptance so far %.4f" % (na / nt))


time.sleep(2.0)
a = job(); a.run(5); reps(a, "A fresh job, cold GPU        ")
a.run(3000); reps(a, "B +3000 transitions (hot)    ")
time.sleep(2.0); reps(a, "C after 2 s idle (cold)      ")
a.run(3000)
d = job(); d.run(5); reps(d, "D fresh job, hot GPU         ")
