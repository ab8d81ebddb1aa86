"""Regenerates the golden vectors under tests/golden/ from the CPU oracle.

    python tests/golden/make_golden.py

The reference (Klara.jl) cannot run here (no Julia) and holds no golden sampler outputs
(F7), so these vectors pin the *build-defined* stream: inputs (x0) and the oracle's outputs (final
state, accept mask, per-chain sums, tuner state) for the parity cases in tests/cases.py.  The GPU
tests compare libklara_hip.so against them bit for bit; the CPU tests check that the oracle still
reproduces them (guards against silent drift of the oracle itself).
swiss.npz / rats.npz are the reference's own data files (data/swiss/*.csv, data/rats/*.csv) as arrays.
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))

import cases  # noqa: E402
import oracle_ffi as O  # noqa: E402
from klara_jl_amd import _lib as L  # noqa: E402


def run_case(name):
    c = cases.make_case(name)
    layout = None                      # the oracle mirrors the product's layout choice (oracle_ffi.default_layout)
    dt = name in cases.DIAGT_CASES     # (these run with the accept mask as their only monitor: no running sums)
    job = O.OracleJob(**cases.oracle_kwargs(c, layout=layout), want_sums=not dt)
    if dt:
        assert job.layout.kind == 3
    st = job.init_state_normal() if c["x0"] is None else job.set_state(c["x0"])
    assert st == 0, (name, st)
    x0 = job.X.copy()
    assert job.run(c["nsteps"]) == 0
    return dict(x0=x0, x=job.X, lt=job.LT, g=job.G, accept=job.accept, sum=job.sum if not dt else np.zeros(0), sumsq=job.sumsq if not dt else np.zeros(0),
                naccept=job.naccept, step=job.step, accepted=job.accepted, proposed=job.proposed,
                totproposed=job.totproposed, da_epsbar=job.da_epsbar, da_hbar=job.da_hbar, layout=np.array([job.layout.kind, job.layout.G, job.layout.E]))


if __name__ == "__main__":
    for name in cases.GOLDEN_CASES:
        out = run_case(name)
        np.savez_compressed(cases.GOLDEN / f"{name}.npz", **out)
        print(name, {k: v.shape for k, v in out.items()})
