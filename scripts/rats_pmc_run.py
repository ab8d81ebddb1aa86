"""The cfg 5 workload alone (rats HMC L=32, pooled tuner, 131,072 chains), for rocprofv3 counter passes."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, cases
import klara_jl_amd as K
from klara_jl_amd import _lib as L
t = cases.rats_target(); n = 131072
x0 = t.least_squares_start()[None, :] + 0.05 * np.random.default_rng(1).standard_normal((n, t.ndims))
e = K.Engine(sampler=L.SAMPLER_HMC, target=t, nchains=n, nsteps=2000, burnin=1000, leapstep=0.02, nleaps=32, steps_per_launch=10,
             tuner=L.TUNER_ACCEPT_RATE, tuner_mode=L.TUNE_POOLED, targetrate=0.65, period=100, monitor=L.MON_SUMMARIES)
e.set_state(x0); e.run(40)
print(e.layout(), e.last_run_ms())
