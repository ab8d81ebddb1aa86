#!/usr/bin/env python3
"""Same-box A/B of library builds (KLARA_HIP_LIB selects the .so) on the headline workload (MALA, lt = -|x|^2, D = 100, 65,536 chains):
steady-state us per transition against the acceptance rate (one stream) for the running-sum modes, the driver-shaped short region
(fresh job, run(5), then 5 x run(20) timed with the host clock) and the default two-stream long run."""
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import klara_jl_amd as K
from klara_jl_amd import _lib as L

if not hasattr(L.load(), "klara_get_launch_modes"):          # an older build of the library
    K.Engine.launch_modes = lambda self: (np.zeros(3, dtype=np.int64), np.zeros(4, dtype=np.int32), np.zeros(4, dtype=np.int64))

tag = os.environ.get("AB_TAG", os.path.basename(os.environ.get("KLARA_HIP_LIB", "default")))
N, D = 65536, 100
neg = K.GaussDiagTarget.negdot(D)
drifts = [float(v) for v in os.environ.get("AB_DRIFTS", "0.9,0.7,0.6,0.5,0.4,0.3,0.1").split(",")]
modes = [int(v) for v in os.environ.get("AB_MODES", "-1,0,1,2").split(",")]     # -1: no save rule
# keep the device busy first (clock ramp)
w = K.Engine(sampler=L.SAMPLER_MALA, target=neg, nchains=N, nsteps=10 ** 7, driftstep=0.9, monitor=0); w.init_state_normal(); w.run(3200)
for mode in modes:
    for h in drifts:
        e = K.Engine(sampler=L.SAMPLER_MALA, target=neg, nchains=N, nsteps=10 ** 7, driftstep=h, monitor=0 if mode < 0 else L.MON_SUMMARIES,
                     sparse_moves=max(mode, 0), nstreams=1)
        e.init_state_normal(); e.run(640)
        _, _, na0, nt0, _ = e.pooled_summaries(with_sums=False)
        c0 = e.launch_modes()[0].copy()
        e.run(640); ms, nl = e.last_run_ms()
        _, _, na, nt, _ = e.pooled_summaries(with_sums=False)
        print(f"[{tag}] steady mode {mode:2d} drift {h:4.2f}: {ms * 1e3 / 640:6.2f} us/transition, acceptance {(na - na0) / (nt - nt0):.4f}, "
              f"launch kinds {tuple(int(v) for v in e.launch_modes()[0] - c0)}", flush=True)
        e.close()
for mode in [m for m in modes if m >= 0]:
    for rep in range(2):
        e = K.Engine(sampler=L.SAMPLER_MALA, target=neg, nchains=N, nsteps=10 ** 7, driftstep=0.9, seed=20260927, monitor=L.MON_SUMMARIES, sparse_moves=mode)
        e.init_state_normal(); e.run(5)
        w.run(3200)
        ts, accs = [], []
        _, _, na0, nt0, _ = e.pooled_summaries(with_sums=False)
        for r in range(5):
            t0 = time.perf_counter(); e.run(20); ts.append((time.perf_counter() - t0) / 20 * 1e6)
            kms, _ = e.last_run_ms()
            _, _, na, nt, _ = e.pooled_summaries(with_sums=False)
            accs.append((na - na0) / (nt - nt0)); na0, nt0 = na, nt
            w.run(320)
        print(f"[{tag}] driver-shaped region mode {mode}: us/transition per repetition {['%.2f' % t for t in ts]} median {np.median(ts):.2f}; "
              f"acceptance per repetition {['%.3f' % a for a in accs]}; launch kinds {tuple(int(v) for v in e.launch_modes()[0])}", flush=True)
        e.close()
for mode in [m for m in modes if m >= 0][:2]:
    e = K.Engine(sampler=L.SAMPLER_MALA, target=neg, nchains=N, nsteps=10 ** 7, driftstep=0.9, seed=20260927, monitor=L.MON_SUMMARIES, sparse_moves=mode)
    e.init_state_normal(); e.run(128)
    ts = []
    for r in range(3):
        t0 = time.perf_counter(); e.run(1024); ts.append((time.perf_counter() - t0) / 1024 * 1e6)
    print(f"[{tag}] default long run (2 streams) mode {mode}: us/transition {['%.2f' % t for t in ts]}; launch kinds {tuple(int(v) for v in e.launch_modes()[0])}", flush=True)
    e.close()
w.close()
