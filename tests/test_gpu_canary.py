"""Memory-safety pass over the raw-buffer addressing.

Every kernel addresses the chains' vectors through buffer resources (`group_window`: num_records = the bytes of the chains that exist) plus
per-lane byte offsets, with KLARA_BUF_OOB offsets for padding lanes: the hardware returns 0 for an out-of-range load and drops the store.
A window one element too large or an offset that is in range when it should not be is silent — unless the neighbours are canaries.
KLARA_DEBUG_CANARY=1 puts 4 KiB of a signalling-NaN pattern before and after EVERY device array (klara_api.hip dalloc_bytes); a stray store
is found when the handle is destroyed (klara_destroy -> KLARA_ERR_STATE -> Engine.close raises) or by klara_selftest_canary, a stray load
brings a NaN into results the parity tests compare bit for bit with the oracle.
"""
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("gpu_required")]
ROOT = Path(__file__).resolve().parent.parent

_PROBE = r'''
import ctypes as C, sys
sys.path.insert(0, "ROOT"); sys.path.insert(0, "ROOT/tests")
import numpy as np
import klara_jl_amd as K
from klara_jl_amd import _lib as L
lib = L.load()
def check(poke=0):
    na, nc = C.c_int64(0), C.c_int64(0)
    L.check(lib.klara_selftest_canary(poke, C.byref(na), C.byref(nc)), "klara_selftest_canary")
    return na.value, nc.value
e = K.Engine(sampler=L.SAMPLER_MALA, target=K.GaussDiagTarget.negdot(100), nchains=4099, nsteps=200, burnin=50, driftstep=0.05, monitor=L.MON_SUMMARIES | L.MON_ACCEPT)
e.init_state_normal(); e.run(200)
na, nc = check()
assert na >= 8 and nc == 0, (na, nc)                          # the job's arrays, all intact after a real run
na, nc = check(poke=1)                                        # an off-by-one store behind the largest array ...
assert nc == 1, nc                                            # ... is seen
assert check() == (na, 0)                                     # (and was repaired: reported once)
e.close()                                                     # intact: no error
# a deliberately broken window: 8 bytes stored directly BEFORE the start of X (what a chain-group window with a negative base would do)
e = K.Engine(sampler=L.SAMPLER_HMC, target=K.GaussDiagTarget.negdot(40), nchains=130, nsteps=20, leapstep=0.1, nleaps=3)
e.init_state_normal(); e.run(20)
x, lt, g = C.c_void_p(), C.c_void_p(), C.c_void_p()
L.check(lib.klara_device_ptrs(e._h, C.byref(x), C.byref(lt), C.byref(g)), "klara_device_ptrs")
hip = C.CDLL("libamdhip64.so")
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
zero = C.c_double(0.0)
assert hip.hipMemcpy(C.c_void_p(x.value - 8), C.byref(zero), 8, 1) == 0
try:
    e.close()
    print("NOT DETECTED")
except K.KlaraError as exc:
    assert exc.status == L.ERR_STATE, exc.status
    print("DETECTED")
'''


def _env():
    env = dict(os.environ, KLARA_DEBUG_CANARY="1")
    return env


def test_canaries_fire_on_a_stray_store():
    r = subprocess.run([sys.executable, "-c", _PROBE.replace("ROOT", str(ROOT))], capture_output=True, text=True, timeout=600, env=_env(), cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-3000:]
    assert r.stdout.strip().endswith("DETECTED") and "NOT DETECTED" not in r.stdout, r.stdout


def test_canaries_are_off_by_default(klib):
    import ctypes as C
    from klara_jl_amd import _lib as L
    if os.environ.get("KLARA_DEBUG_CANARY") == "1":
        pytest.skip("this session runs with the canaries on")
    assert klib.klara_selftest_canary(0, None, None) == L.ERR_STATE


def test_parity_sweeps_and_full_size_workloads_between_canaries():
    """The configuration sweeps — random configurations over the whole accepted space, every dimension on both Gaussian layouts, every pairs-per-lane
    count, tiny and ragged chain counts — and the BASELINE workloads at their stated sizes, once more with every device array between canaries:
    all green, every handle's canaries intact when it is destroyed, nothing damaged among what is alive at the end (tests/conftest.py)."""
    if os.environ.get("KLARA_DEBUG_CANARY") == "1":
        pytest.skip("already inside the canary run")
    sel = ("random_configurations or dimension_sweep or every_pairs_per_lane or tiny_jobs or chain_count_sweep or other_unit_counts or full_size "
           "or cfg4 or cfg5 or test_parity_with_oracle or parity_pair_transposed_layout or launch_splitting or history_ring")
    r = subprocess.run([sys.executable, "-m", "pytest", str(ROOT / "tests"), "-m", "gpu", "-x", "-q", "-k", sel, "-p", "no:cacheprovider"],
                       capture_output=True, text=True, timeout=3000, env=_env(), cwd=str(ROOT))
    tail = r.stdout[-1500:]
    assert r.returncode == 0, tail + r.stderr[-1500:]
    assert "[klara canary]" in r.stdout and " 0 with damaged canaries" in r.stdout, tail
    assert " passed" in tail and "failed" not in tail, tail
