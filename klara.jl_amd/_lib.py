"""ctypes binding of libklara_hip.so (include/klara_hip.h).

The library is the product: there is no Python/NumPy/CPU implementation of the transition path behind
this module.  If the shared object is missing, `load()` raises; if no HIP device is present,
`klara_create` returns KLARA_ERR_HIP and `KlaraError` is raised.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "lib" / "libklara_hip.so"

KLARA_ABI_VERSION = 6
DEFAULT_STEPS_PER_LAUNCH = 32     # KLARA_DEFAULT_STEPS_PER_LAUNCH
DEFAULT_STEPS_PER_LAUNCH_SLICE = 128   # KLARA_DEFAULT_STEPS_PER_LAUNCH_SLICE (the slice-sampler jobs the free-running kernel serves)
LOGIT_MAX_LDS_DOUBLES = 18432     # KLARA_LOGIT_MAX_LDS_DOUBLES

# klara_status
OK, ERR_INVALID_ARG, ERR_NONFINITE_INIT, ERR_HIP, ERR_NOMEM, ERR_UNSUPPORTED, ERR_STATE, ERR_SLICE_STUCK, ERR_COMPILE = range(9)
# klara_sampler
SAMPLER_MH, SAMPLER_MALA, SAMPLER_HMC, SAMPLER_SLICE = range(4)
# klara_target
TARGET_GAUSS_DIAG, TARGET_GAUSS_DENSE, TARGET_LOGISTIC, TARGET_HIER_NORMAL, TARGET_CUSTOM = range(5)
# klara_tuner / mode
TUNER_VANILLA, TUNER_ACCEPT_RATE, TUNER_DUAL_AVERAGING = 0, 1, 2
TUNE_PER_CHAIN, TUNE_POOLED = 0, 1
MON_ACCEPT, MON_HISTORY, MON_SUMMARIES, MON_HIST_LT, MON_HIST_GRAD, MON_HIST_LLLP = 0x1, 0x2, 0x4, 0x8, 0x10, 0x20

_dp = C.POINTER(C.c_double)


class KlaraDesc(C.Structure):
    """struct klara_desc (include/klara_hip.h) — field order and types must match exactly."""

    _fields_ = [
        ("struct_size", C.c_uint32), ("abi_version", C.c_uint32),
        ("sampler", C.c_int32), ("target", C.c_int32), ("tuner", C.c_int32), ("tuner_mode", C.c_int32),
        ("nchains", C.c_int64), ("chain_offset", C.c_int64), ("ndims", C.c_int32), ("device", C.c_int32),
        ("mh_sigma", _dp), ("driftstep", C.c_double), ("leapstep", C.c_double),
        ("nleaps", C.c_int32), ("slice_stepout", C.c_int32), ("slice_widths", _dp),
        ("targetrate", C.c_double), ("score_k", C.c_double), ("period", C.c_int32), ("verbose", C.c_int32),
        ("da_nadapt", C.c_int64), ("da_eps0bar", C.c_double), ("da_h0bar", C.c_double), ("da_gamma", C.c_double),
        ("da_kappa", C.c_double), ("da_t0", C.c_int32), ("tuner_score", C.c_int32),
        ("nsteps", C.c_int64), ("burnin", C.c_int64), ("thinning", C.c_int64),
        ("gauss_w", _dp), ("gauss_mu", _dp), ("gauss_const", C.c_double), ("gauss_prec", _dp),
        ("logit_X", _dp), ("logit_y", _dp), ("logit_ndata", C.c_int32), ("nstreams", C.c_int32),
        ("logit_lambda", C.c_double),
        ("hier_Y", _dp), ("hier_xc", _dp), ("hier_nunits", C.c_int32), ("hier_ntimes", C.c_int32),
        ("hier_prior_prec", C.c_double), ("hier_gamma_a", C.c_double), ("hier_gamma_b", C.c_double),
        ("custom_src", C.c_char_p), ("custom_data", _dp), ("custom_ndata", C.c_int64), ("bm_batchlen", C.c_int64),
        ("hist_ring_cols", C.c_int64), ("acov_maxlag", C.c_int32), ("sparse_moves", C.c_int32),
        ("seed", C.c_uint64), ("monitor", C.c_uint32), ("steps_per_launch", C.c_int32),
        ("stream", C.c_void_p),
    ]


class KlaraError(RuntimeError):
    def __init__(self, status: int, where: str):
        self.status = int(status)
        msg = _strerror(status)
        self.log = ""
        if self.status == ERR_COMPILE:                 # the compiler's message for a user-defined target
            try:
                self.log = load().klara_compile_log().decode(errors="replace")
            except Exception:
                pass
        super().__init__(f"{where}: klara_status {int(status)} ({msg})" + (("\n" + self.log) if self.log else ""))


# every symbol include/klara_hip.h declares (tests check the .so exports all of them)
EXPORTS = [
    "klara_create", "klara_destroy", "klara_set_state", "klara_init_state_normal", "klara_run",
    "klara_run_async", "klara_synchronize", "klara_reset", "klara_stream_key", "klara_get_state", "klara_get_accept_mask", "klara_get_accept_rows",
    "klara_get_accept_counts", "klara_get_chain_sums", "klara_get_pooled_summaries", "klara_get_chain",
    "klara_get_chain_fields", "klara_get_chain_likelihood_prior", "klara_get_chain_mcvar", "klara_get_chain_mcvar_ipse", "klara_get_chain_acov_mcvar", "klara_saved_steps", "klara_get_chain_bm", "klara_get_tune", "klara_get_dual_averaging", "klara_last_run_ms", "klara_device_ptrs", "klara_get_layout", "klara_get_launch_modes", "klara_get_kernel_attributes", "klara_get_shader_clock",
    "klara_selftest_rocrand_blocks", "klara_selftest_math", "klara_selftest_normal_tail", "klara_selftest_transition_normals", "klara_selftest_mfma_f64", "klara_selftest_mfma_f64_4x4x4", "klara_strerror",
    "klara_comm_unique_id", "klara_comm_init", "klara_comm_info", "klara_comm_destroy", "klara_gather_summaries", "klara_gather_moments",
    "klara_check_custom_target", "klara_compile_log", "klara_selftest_plan", "klara_selftest_canary", "klara_abi_version",
]

_lib = None


def load() -> C.CDLL:
    """Load libklara_hip.so (built in-tree by __graft_entry__.build() / klara.jl_amd/csrc/Makefile)."""
    global _lib
    if _lib is not None:
        return _lib
    path = Path(os.environ.get("KLARA_HIP_LIB", LIB_PATH))
    if not path.exists():
        raise FileNotFoundError(
            f"{path} not found: build it first (python -c 'import __graft_entry__ as g; g.build()' "
            f"or make -C klara.jl_amd/csrc). There is no CPU fallback for the transition path.")
    lib = C.CDLL(str(path))
    H = C.c_void_p
    i64p, u64p = C.POINTER(C.c_int64), C.POINTER(C.c_uint64)
    sig = {
        "klara_create": [C.POINTER(KlaraDesc), C.POINTER(H)],
        "klara_destroy": [H],
        "klara_set_state": [H, C.c_void_p],
        "klara_init_state_normal": [H],
        "klara_run": [H, C.c_int64],
        "klara_run_async": [H, C.c_int64],
        "klara_synchronize": [H],
        "klara_reset": [H, C.c_void_p],
        "klara_stream_key": [H, u64p, u64p],
        "klara_get_state": [H, C.c_void_p, C.c_void_p, C.c_void_p],
        "klara_get_accept_mask": [H, C.c_void_p, C.c_int64, i64p],
        "klara_get_accept_rows": [H, C.c_int64, C.c_int64, C.c_void_p],
        "klara_get_accept_counts": [H, C.c_void_p, u64p],
        "klara_get_chain_sums": [H, C.c_void_p, C.c_void_p, i64p],
        "klara_get_pooled_summaries": [H, C.c_void_p, C.c_void_p, u64p, u64p, i64p],
        "klara_get_chain": [H, C.c_int64, C.c_void_p, C.c_int64, i64p],
        "klara_get_chain_fields": [H, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, i64p],
        "klara_get_chain_likelihood_prior": [H, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, i64p],
        "klara_get_chain_mcvar": [H, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p],
        "klara_get_tune": [H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
        "klara_get_dual_averaging": [H, C.c_void_p, C.c_void_p],
        "klara_last_run_ms": [H, C.POINTER(C.c_double), i64p],
        "klara_device_ptrs": [H, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)],
        "klara_get_layout": [H, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)],
        "klara_get_launch_modes": [H, C.c_void_p, C.c_void_p, C.c_void_p],
        "klara_get_kernel_attributes": [H, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)],
        "klara_get_shader_clock": [H, C.POINTER(C.c_double)],
        "klara_selftest_rocrand_blocks": [C.c_int32, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int32, C.c_void_p],
        "klara_selftest_math": [C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p],
        "klara_selftest_normal_tail": [C.c_int32, C.c_uint64, C.c_uint64, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p],
        "klara_selftest_transition_normals": [C.c_int32, C.c_uint64, C.c_uint64, C.c_int64, C.c_uint64, C.c_int32, C.c_void_p, C.c_void_p],
        "klara_selftest_mfma_f64": [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
        "klara_selftest_mfma_f64_4x4x4": [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
        "klara_comm_unique_id": [C.c_void_p],
        "klara_comm_init": [C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.c_void_p, C.c_int32],
        "klara_comm_info": [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)],
        "klara_comm_destroy": [C.c_void_p],
        "klara_gather_summaries": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                   C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)],
        "klara_gather_moments": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                 C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)],
        "klara_selftest_canary": [C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)],
        "klara_check_custom_target": [C.c_char_p, C.c_int32, C.c_int32],
        "klara_get_chain_bm": [C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)],
        "klara_get_chain_acov_mcvar": [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)],
        "klara_get_chain_mcvar_ipse": [C.c_void_p, C.c_int64, C.c_void_p],
        "klara_saved_steps": [C.c_void_p, C.POINTER(C.c_int64)],
        "klara_selftest_plan": [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.POINTER(C.c_int64)],
    }
    for name, argtypes in sig.items():
        if "KLARA_HIP_LIB" in os.environ and not hasattr(lib, name):
            continue            # (same-box A/B against an older build of the library, scripts/ab_builds.py: it may lack the newest entry points)
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = C.c_int
    lib.klara_strerror.argtypes = [C.c_int]
    lib.klara_strerror.restype = C.c_char_p
    lib.klara_compile_log.argtypes = []
    lib.klara_compile_log.restype = C.c_char_p
    lib.klara_abi_version.argtypes = []
    lib.klara_abi_version.restype = C.c_int32
    got = int(lib.klara_abi_version())
    if got != KLARA_ABI_VERSION:
        # An older build differs in its random stream (and, before version 4, in what klara_desc.sparse_moves = 0 means), so it is never
        # accepted silently: only the same-box A/B scripts ask for it (KLARA_HIP_LIB=<older build> KLARA_ALLOW_ABI_MISMATCH=1), and only
        # for versions whose klara_desc layout this binding can fill (3 .. current).
        if os.environ.get("KLARA_ALLOW_ABI_MISMATCH") != "1" or "KLARA_HIP_LIB" not in os.environ or not (3 <= got <= KLARA_ABI_VERSION):
            raise RuntimeError(f"libklara_hip.so ABI version mismatch: library {got}, binding {KLARA_ABI_VERSION}")
        import warnings
        warnings.warn(f"{path}: ABI version {got} (binding {KLARA_ABI_VERSION}) accepted because KLARA_ALLOW_ABI_MISMATCH=1 — "
                      "its random stream and launch defaults differ from this version's", RuntimeWarning)
        lib._klara_abi_override = got          # descriptors for THIS library carry its own version (klara_create checks it)
    _lib = lib
    return lib


def _strerror(status: int) -> str:
    try:
        return load().klara_strerror(int(status)).decode()
    except Exception:  # library missing: still give a readable message
        return "libklara_hip.so unavailable"


def check(status: int, where: str) -> None:
    if status != OK:
        raise KlaraError(status, where)
