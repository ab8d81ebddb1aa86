import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
if str(ROOT / "tests") not in sys.path:
    sys.path.insert(0, str(ROOT / "tests"))


# run-time compiled code objects of user-defined targets: cached inside the repository tree (git-ignored), never in $HOME
os.environ.setdefault("KLARA_JIT_CACHE_DIR", str(ROOT / "build" / "jit_cache"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """Make sure the in-tree shared objects are current (a no-op `make` when they are): libklara_hip.so
    (hipcc, gfx950 — cross-compiles without a GPU) and the CPU oracle (gcc)."""
    import shutil
    import subprocess
    if shutil.which("make") is None:
        return
    if shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"):
        subprocess.run(["make", "-C", str(ROOT / "klara.jl_amd" / "csrc"), "-j8"], capture_output=True)
    subprocess.run(["make", "-C", str(ROOT / "oracle")], capture_output=True)


def _gpu_available() -> bool:
    try:
        import torch
        return bool(torch.cuda.is_available())
    except Exception:
        return False


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): built on demand with gcc, loaded via ctypes."""
    import oracle_ffi
    return oracle_ffi.load()


@pytest.fixture(scope="session")
def klib():
    """libklara_hip.so through the product's own binding; must already be built (build() does it)."""
    import klara_jl_amd
    return klara_jl_amd._lib.load()


@pytest.fixture(scope="session")
def gpu_required():
    if not _gpu_available():
        pytest.fail("-m gpu tests need a GPU: torch.cuda.is_available() is False")


def pytest_sessionfinish(session, exitstatus):
    """KLARA_DEBUG_CANARY=1 (tests/test_gpu_canary.py runs the sweeps that way): whatever is still alive at the end of the session must have
    intact canaries too — a damaged one fails the run."""
    if os.environ.get("KLARA_DEBUG_CANARY") != "1" or not _gpu_available():
        return
    import ctypes as C
    import klara_jl_amd
    lib = klara_jl_amd._lib.load()
    na, nc = C.c_int64(0), C.c_int64(0)
    st = lib.klara_selftest_canary(0, C.byref(na), C.byref(nc))
    print(f"\n[klara canary] {na.value} live device arrays checked at session end, {nc.value} with damaged canaries (status {st})")
    if st != 0 or nc.value != 0:
        session.exitstatus = 1
