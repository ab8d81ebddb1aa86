"""ctypes access to oracle/libklara_oracle.so — TEST INFRASTRUCTURE (never imported by the product).

`run_oracle(**cfg)` executes the CPU restatement of the reference transition path on the same
descriptor the product's C ABI takes and returns everything the parity tests compare.
"""
from __future__ import annotations

import ctypes as C
import hashlib
import os
import subprocess
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
SO = ROOT / "oracle" / "libklara_oracle.so"

import klara_jl_amd  # noqa: E402  (struct definition + enums only)
from klara_jl_amd import _lib as L  # noqa: E402


class KoLayout(C.Structure):
    _fields_ = [("kind", C.c_int32), ("G", C.c_int32), ("E", C.c_int32)]


_lib = None


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    srcs = [ROOT / "oracle" / "klara_oracle.c", ROOT / "include" / "klara_hip.h",
            ROOT / "klara.jl_amd" / "csrc" / "detmath.h"]
    if not SO.exists() or any(s.stat().st_mtime > SO.stat().st_mtime for s in srcs):
        r = subprocess.run(["make", "-C", str(ROOT / "oracle")], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("oracle build failed:\n" + r.stdout + r.stderr)
    lib = C.CDLL(str(SO))
    vp = C.c_void_p
    lib.ko_init.argtypes = [C.POINTER(L.KlaraDesc), C.POINTER(KoLayout)] + [vp] * 9
    lib.ko_init.restype = C.c_int
    lib.ko_init_state_normal.argtypes = [C.POINTER(L.KlaraDesc), vp]
    lib.ko_init_state_normal.restype = None
    lib.ko_run.argtypes = [C.POINTER(L.KlaraDesc), C.POINTER(KoLayout)] + [vp] * 7 + [C.c_int64, C.c_int64] + [vp] * 5 + [C.c_int64, vp, vp, vp, vp, vp]
    lib.ko_run.restype = C.c_int
    lib.ko_philox_block.argtypes = [vp, vp, vp]
    lib.ko_stream_block.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, vp]
    lib.ko_math.argtypes = [C.c_int, C.c_int64, vp, vp, vp]
    lib.ko_u52.argtypes = [C.c_uint32, C.c_uint32]
    lib.ko_u44.argtypes = [C.c_uint32, C.c_uint32]; lib.ko_u44.restype = C.c_double
    lib.ko_slice_draws.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int32, vp]; lib.ko_slice_draws.restype = None
    lib.ko_normal_pairs_w.argtypes = [C.c_int64, vp, vp]; lib.ko_normal_pairs_w.restype = None
    lib.ko_transition_normals.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_int32, vp, vp]; lib.ko_transition_normals.restype = None
    lib.ko_normal_tail.argtypes = [C.c_uint64, C.c_uint64, C.c_int64, C.c_int64, C.c_int32, vp, vp, vp]
    lib.ko_normal_tail.restype = None
    lib.ko_u52.restype = C.c_double
    lib.ko_eval_target.argtypes = [C.POINTER(L.KlaraDesc), C.POINTER(KoLayout), vp, vp, vp]
    lib.ko_eval_target.restype = C.c_int
    lib.ko_bm_close.argtypes = [vp, vp, vp, vp, C.c_int64, C.c_int64, C.c_int64]
    lib.ko_bm_close.restype = None
    lib.ko_set_custom_target.argtypes = [vp, vp]
    lib.ko_set_custom_target.restype = None
    lib.ko_set_custom_pair_target.argtypes = [vp]
    lib.ko_set_custom_pair_target.restype = None
    lib.ko_set_literal.argtypes = [C.c_int]        # literal Julia arithmetic (tests/test_literal_arithmetic.py only)
    lib.ko_set_literal.restype = None
    lib.ko_get_literal.argtypes = []
    lib.ko_get_literal.restype = C.c_int
    for name in ("ko_logistic",):
        getattr(lib, name).argtypes = [C.c_double] * 5
        getattr(lib, name).restype = C.c_double
    for name in ("ko_logistic_rate_score", "ko_erf_rate_score"):
        getattr(lib, name).argtypes = [C.c_double] * 2
        getattr(lib, name).restype = C.c_double
    _lib = lib
    return lib


_user_libs = {}


def compile_user_target(src: str, ndims: int):
    """Host form of a user-defined target (KLARA_TARGET_CUSTOM): the same C text the product hands to hiprtc, compiled by
    gcc with the same arithmetic contract (-ffp-contract=off, detmath.h for kd_*).  Returns (lib, lt_ptr, grad_ptr|None)."""
    key = hashlib.sha1(f"v2 {ndims}\n{src}".encode()).hexdigest()[:16]
    if key in _user_libs:
        return _user_libs[key]
    out = ROOT / "oracle" / "_user"
    out.mkdir(exist_ok=True)
    so, c = out / f"user_{key}.so", out / f"user_{key}.c"
    if not so.exists():
        # (a likelihood + prior source without gradient closures — an MH / slice job — gets no composed gradient, as on the device)
        nograd = "#define KLARA_CUSTOM_NOGRAD 1\n" if ("KLARA_USER_LIKELIHOOD_PRIOR" in src and "klara_user_gradloglikelihood" not in src) else ""
        c.write_text(f'#include "detmath.h"\n#define KLARA_D {int(ndims)}\n#define KLARA_USER_FN\n{nograd}#line 1 "klara_user_target"\n{src}\n'
                     '#line 1 "klara_custom_glue"\n#include "klara_custom_compose.h"\n')      # (likelihood + prior form: same composition as the device)
        r = subprocess.run(["gcc", "-O2", "-std=gnu11", "-ffp-contract=off", "-fPIC", "-shared", "-I", str(ROOT / "klara.jl_amd" / "csrc"),
                            "-o", str(so), str(c), "-lm"], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("user target did not compile on the host:\n" + r.stderr)
    lib = C.CDLL(str(so))
    if "KLARA_USER_PAIR_TARGET" in src and "KLARA_PAIR_AS_WHOLE" not in src:        # pair closure (klara_user_pair): third entry; no whole-vector closures
        _user_libs[key] = (lib, None, None, C.cast(lib.klara_user_pair, C.c_void_p))
        return _user_libs[key]
    lt = C.cast(lib.klara_user_logtarget, C.c_void_p)
    grad = C.cast(lib.klara_user_gradlogtarget, C.c_void_p) if hasattr(lib, "klara_user_gradlogtarget") else None
    _user_libs[key] = (lib, lt, grad, None)
    return _user_libs[key]


DIAGT_NP_MENU = (2, 3, 4, 5, 6, 7, 8)      # klara_launch.h KLARA_DIAGT_NP_MENU_DO
DIAGT_Q = 8


def split_dense_layout(d: int):
    """(6, wavefronts per tile, elements per lane and wavefront) of the workgroup-split dense layout: klara_launch.h klara_split_new / klara_split_waves —
    16, 24 or 32 elements (4 / 6 / 8 row tiles) per lane and wavefront: whichever puts the fewest wavefronts on a tile (24: 257 .. 384 and 513 .. 768 dimensions; 32: 385 .. 512 and 769 .. 1024)."""
    mt = (d + 15) // 16
    new, wbest = 16, 4 * ((mt + 15) // 16)
    for n in (24, 32):                     # the fewest wavefronts on a tile, the smaller element count on a tie
        wn = 4 * ((mt + n - 1) // n)
        if wn < wbest and wn <= 8:
            new, wbest = n, wn
    if os.environ.get("KLARA_SPLIT_NEW") in ("16", "24", "32"):
        new = int(os.environ["KLARA_SPLIT_NEW"])
    w = 4 * ((mt + new - 1) // new)
    if os.environ.get("KLARA_SPLIT_W", "").isdigit() and int(os.environ["KLARA_SPLIT_W"]) >= w and int(os.environ["KLARA_SPLIT_W"]) % 4 == 0 and int(os.environ["KLARA_SPLIT_W"]) <= 16:
        w = int(os.environ["KLARA_SPLIT_W"])
    return (6, w, new)


def default_layout(target_kind: int, ndims: int, ndata: int = 0, sampler=None, tuner=0, tuner_mode=0, verbose=False,
                   hier_nunits: int = 0, hier_ntimes: int = 0, summaries: bool = True, sparse_moves: bool = False, pair_form: bool = False, custom_rows: int = 2):
    """Mirror of the product's layout choice (klara_get_layout reports the real one on the GPU box).  `sampler`, `tuner`,
    `tuner_mode` and `verbose` are only needed to recognise the pair-transposed layout (kind 3): diagonal Gaussian,
    MH / MALA / HMC, even D <= 128, Vanilla or AcceptanceRate tuner (klara_api.hip diagt_eligible)."""
    d = int(ndims)
    plain = sampler is not None      # (name kept from when the layout excluded tuned jobs)
    if (target_kind == L.TARGET_GAUSS_DIAG and sampler is not None and plain
            and 17 <= d <= 1024 and os.environ.get("KLARA_LAYOUT_KIND", "3") != "0" and "KLARA_LAYOUT_E" not in os.environ):
        q = 8 if d <= 128 else (16 if d <= 256 else (32 if d <= 512 else 64))          # lanes per chain (klara_api.hip select_layout)
        # (untuned MH / MALA up to D = 104 also run on 4-lane kernels: they sum in this 8-lane order, klara_diagt.h)
        return (3, q, 2 * ((d + 2 * q - 1) // (2 * q)))
    if (target_kind == L.TARGET_HIER_NORMAL and sampler in (L.SAMPLER_MH, L.SAMPLER_MALA, L.SAMPLER_HMC)
            and 9 <= hier_nunits <= 32 and os.environ.get("KLARA_LAYOUT_KIND", "4") != "0"):
        return (4, 8, 8)              # klara_hiert.h: 8 lanes per chain, 4 units per lane
    if target_kind == L.TARGET_CUSTOM and pair_form:     # pair closure: the pair-transposed layout (klara_api.hip select_layout)
        q = 8 if d <= 128 else (16 if d <= 256 else (32 if d <= 512 else 64))
        return (3, q, 2 * ((d + 2 * q - 1) // (2 * q)))
    if target_kind == L.TARGET_CUSTOM and d > 32 and os.environ.get("KLARA_CUSTOM_LANES", "0") != "1":
        # whole-vector closure staged through LDS (klara_api.hip custom_layout): G lanes x E = 2 ceil(D / 2G) <= 16 elements, a workgroup's
        # rows (4 wavefronts x 64 / G chains, `custom_rows` vectors each) within 56 KB
        g = int(os.environ.get("KLARA_CUSTOM_LANES", "0")) or 4
        rows = custom_rows
        stride = rows * ((d + 1) & ~1) + 2
        stride += 2 if (2 * stride) % 64 == 0 else 0
        while g < 64 and (d + 2 * g - 1) // (2 * g) > 8:          # at most 16 elements per lane (round 6: up to 64 lanes, D <= 1024)
            g *= 2
        wpb = int(os.environ.get("KLARA_CUSTOM_WPB", "0")) or (2 if (g >= 8 and 4 * (64 // g) * stride * 8 > 57344) else 4)   # wavefronts per workgroup
        while g < 64 and wpb * (64 // g) * stride * 8 > 57344:
            g *= 2
        return (0, g, 2 * ((d + 2 * g - 1) // (2 * g)))
    if target_kind == L.TARGET_CUSTOM:        # one chain per lane, pow2ceil(D) elements in registers (klara_custom.h)
        e = 2
        while e < d:
            e *= 2
        return (0, 1, e)
    if (target_kind == L.TARGET_GAUSS_DENSE and d <= 1024 and "KLARA_DENSE_NO_SPLIT" not in os.environ
            and (d > 256 or os.environ.get("KLARA_DENSE_SPLIT", "0") not in ("", "0"))):
        # round 6: a workgroup of 4, 8, 12 or 16 wavefronts per tile of 16 chains, the ceil(D / 16) row tiles dealt evenly (klara_dense_split.h)
        return split_dense_layout(d)
    if (target_kind == L.TARGET_GAUSS_DENSE and 128 < d <= 256 and "KLARA_DENSE_NO_STREAM" not in os.environ
            and not (sampler == L.SAMPLER_SLICE and "KLARA_DENSE_SLICE_NO_STREAM" in os.environ)):      # every sampler to D = 256: still the matrix cores, P streamed (klara_dense_big.h; round 5: the slice sampler too)
        return (1, 4, 8 * ((d + 31) // 32))
    if target_kind == L.TARGET_GAUSS_DENSE and d > 128:     # beyond the matrix-core layouts: the run-time compiled closure form
        return (0, 1, 256)
    if target_kind == L.TARGET_GAUSS_DENSE:
        ne = 8 if d <= 32 else 16 if d <= 64 else 25 if d <= 100 else 32
        return (1, 4, ne)
    if (target_kind == L.TARGET_LOGISTIC and (16 < d <= 256 or (8 < d <= 16 and ndata * 17 > 18432)) and sampler is not None
            and "KLARA_LOGIT_NO_MFMA" not in os.environ):      # round 6: on the matrix cores to 256 parameters (klara_logit_mfma.h): kind 5, 4 lanes per chain, NE = 8 ceil(D / 32)
        return (5, 4, 8 * ((d + 31) // 32))
    if target_kind == L.TARGET_LOGISTIC and (d > 16 or (d > 8 and ndata * 17 > 18432)):     # beyond 16 parameters (or rows that do not fit the LDS): the run-time compiled closure form, one chain per lane
        e = 16
        while e < d:
            e *= 2
        return (0, 1, e)
    if target_kind == L.TARGET_LOGISTIC:
        e = 2 if d <= 2 else 4 if d <= 4 else 8 if d <= 8 else 16
        return (2, 4, e) if ndata >= 64 else (0, 1, e)
    def p2(v):
        g = 1
        while g < v:
            g *= 2
        return g
    e = 4 if d <= 256 else 8
    if d <= 128 and p2((d + 1) // 2) * 2 < p2((d + 3) // 4) * 4:
        e = 2
    return (0, p2((d + e - 1) // e), e)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class OracleJob:
    """Holds a descriptor + state and steps it with ko_run."""

    def __init__(self, *, sampler, target_kind, nchains, ndims, nsteps, burnin=0, thinning=1,
                 mh_sigma=None, driftstep=1.0, leapstep=0.1, nleaps=10, slice_widths=None, slice_stepout=True,
                 tuner=0, tuner_mode=0, targetrate=0.0, score_k=7.0, period=100, verbose=False,
                 da_nadapt=0, da_eps0bar=1.0, da_h0bar=0.0, da_gamma=0.05, da_t0=10, da_kappa=0.75, tuner_score=0,
                 seed=20260927, chain_offset=0, gauss_w=None, gauss_mu=None, gauss_const=0.0, gauss_prec=None,
                 logit_X=None, logit_y=None, logit_lambda=100.0, hier_Y=None, hier_xc=None, hier_prior_prec=1e-4,
                 hier_gamma_a=1e-3, hier_gamma_b=1e-3, custom_src=None, custom_data=None, layout=None,
                 want_accept=True, want_sums=True, want_hist=False, sparse_moves=False):
        self.lib = load()
        self.N, self.D = int(nchains), int(ndims)
        d = L.KlaraDesc()
        d.struct_size = C.sizeof(L.KlaraDesc); d.abi_version = L.KLARA_ABI_VERSION
        d.sampler, d.target, d.tuner, d.tuner_mode = int(sampler), int(target_kind), int(tuner), int(tuner_mode)
        d.nchains, d.chain_offset, d.ndims = self.N, int(chain_offset), self.D
        self._keep = []

        def ptr(a, n=None):
            if a is None:
                return None
            a = _f64(np.broadcast_to(_f64(a).ravel(), (n,)) if n is not None else a)
            self._keep.append(a)
            return a.ctypes.data_as(C.POINTER(C.c_double))

        d.mh_sigma = ptr(mh_sigma, self.D)
        d.slice_widths = ptr(slice_widths, self.D)
        d.driftstep, d.leapstep, d.nleaps, d.slice_stepout = float(driftstep), float(leapstep), int(nleaps), int(bool(slice_stepout))
        d.targetrate, d.score_k, d.period, d.verbose = float(targetrate), float(score_k), int(period), int(bool(verbose))
        d.nsteps, d.burnin, d.thinning = int(nsteps), int(burnin), int(thinning)
        d.da_nadapt, d.da_eps0bar, d.da_h0bar = int(da_nadapt), float(da_eps0bar), float(da_h0bar)
        d.da_gamma, d.da_kappa, d.da_t0 = float(da_gamma), float(da_kappa), int(da_t0)
        d.tuner_score = int(tuner_score)
        d.gauss_w, d.gauss_mu, d.gauss_const = ptr(gauss_w, self.D), ptr(gauss_mu, self.D), float(gauss_const)
        d.gauss_prec = ptr(gauss_prec)
        d.logit_X, d.logit_y = ptr(logit_X), ptr(logit_y)
        d.logit_ndata = 0 if logit_y is None else int(np.size(logit_y))
        d.logit_lambda = float(logit_lambda)
        d.hier_Y, d.hier_xc = ptr(hier_Y), ptr(hier_xc)
        if hier_Y is not None:
            d.hier_nunits, d.hier_ntimes = int(np.shape(hier_Y)[0]), int(np.shape(hier_Y)[1])
        d.hier_prior_prec, d.hier_gamma_a, d.hier_gamma_b = float(hier_prior_prec), float(hier_gamma_a), float(hier_gamma_b)
        self._user = None
        if custom_src is not None:
            # (a pair closure the pair-transposed kernels do not serve runs as a whole-vector closure: klara_api.hip pair_as_whole, klara_custom_compose.h)
            if "KLARA_USER_PAIR_TARGET" in custom_src and (self.D < 17 or (int(sampler) == L.SAMPLER_SLICE and "KLARA_PAIR_SLICE_AS_WHOLE" in os.environ)):
                custom_src = "#define KLARA_PAIR_AS_WHOLE 1\n" + custom_src
            self._user = compile_user_target(custom_src, self.D)
            if custom_data is not None and np.size(custom_data):
                d.custom_data = ptr(custom_data); d.custom_ndata = int(np.size(custom_data))
        d.seed = int(seed)
        self.desc = d
        self._seed0, self.epoch = int(seed), 0
        k, g, e = layout if layout is not None else default_layout(int(target_kind), self.D, int(d.logit_ndata), sampler=int(sampler),
                                                                   tuner=int(tuner), tuner_mode=int(tuner_mode), verbose=bool(verbose),
                                                                   hier_nunits=int(d.hier_nunits), hier_ntimes=int(d.hier_ntimes),
                                                                   summaries=bool(want_sums), sparse_moves=bool(sparse_moves),
                                                                   pair_form=bool(custom_src is not None and "KLARA_USER_PAIR_TARGET" in custom_src and "KLARA_PAIR_AS_WHOLE" not in custom_src),
                                                                   custom_rows=3 if (custom_src is not None and "KLARA_USER_LIKELIHOOD_PRIOR" in custom_src) else 2)
        self.layout = KoLayout(k, g, e)
        nt = 1 if tuner_mode == L.TUNE_POOLED else self.N
        self.X = np.zeros((self.N, self.D)); self.G = np.zeros((self.N, self.D)); self.LT = np.zeros(self.N)
        self.step = np.zeros(nt); self.accepted = np.zeros(nt, np.int64)
        self.proposed = np.zeros(nt, np.int64); self.totproposed = np.zeros(nt, np.int64)
        self.da_epsbar = np.zeros(nt); self.da_hbar = np.zeros(nt)
        self.t = 0
        self.naccept = np.zeros(self.N, np.uint64)
        # running sums in sojourn form (klara_oracle.c ko_run): folded parts + the saved steps held at the current state
        self._sum = np.zeros((self.N, self.D)) if want_sums else None
        self._sumsq = np.zeros((self.N, self.D)) if want_sums else None
        self.held = np.zeros(self.N, np.int64)
        self.hist_cols = (int(nsteps) - int(burnin) - 1) // int(thinning) + 1
        self.hist = np.zeros((self.hist_cols, self.N, self.D)) if want_hist else None
        self.hist_lt = np.zeros((self.hist_cols, self.N)) if want_hist else None
        self.hist_g = np.zeros((self.hist_cols, self.N, self.D)) if want_hist else None
        self.want_accept = want_accept
        self.accept = np.zeros((0, self.N), np.uint8)

    def _p(self, a):
        return None if a is None else a.ctypes.data

    @property
    def sum(self):
        """sum of the saved values per (chain, dimension): folded part + held * x (the view klara_get_chain_sums returns)"""
        return None if self._sum is None else self._sum + self.held[:, None].astype(np.float64) * self.X

    @property
    def sumsq(self):
        return None if self._sumsq is None else self._sumsq + self.held[:, None].astype(np.float64) * (self.X * self.X)

    def _bind_user(self):
        if self._user is not None:             # (process-global in the oracle: bound before every call that evaluates the target)
            self.lib.ko_set_custom_target(self._user[1], self._user[2])
            self.lib.ko_set_custom_pair_target(self._user[3])

    def set_state(self, x) -> int:
        self.X[...] = _f64(x).reshape(self.N, self.D)
        return self._init()

    def reset(self, x=None) -> int:
        """reset(job[, x]): the job moves on to the next Philox key (include/klara_hip.h klara_reset), everything else rewinds"""
        self.epoch = getattr(self, "epoch", 0) + 1
        if not hasattr(self, "_seed0"):
            self._seed0 = int(self.desc.seed)
        self.desc.seed = (self._seed0 + self.epoch * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        if x is not None:
            self.X[...] = _f64(x).reshape(self.N, self.D)
        return self._init()

    def init_state_normal(self) -> int:
        self.lib.ko_init_state_normal(C.byref(self.desc), self.X.ctypes.data)
        return self._init()

    def _init(self) -> int:
        self.G[...] = 0.0
        self.t = 0
        self.naccept[...] = 0
        if self._sum is not None:
            self._sum[...] = 0.0; self._sumsq[...] = 0.0
        self.held[...] = 0
        self.accept = np.zeros((0, self.N), np.uint8)
        self._bind_user()
        return self.lib.ko_init(C.byref(self.desc), C.byref(self.layout), self._p(self.X), self._p(self.G),
                                self._p(self.LT), self._p(self.step), self._p(self.accepted),
                                self._p(self.proposed), self._p(self.totproposed), self._p(self.da_epsbar),
                                self._p(self.da_hbar))

    def run(self, nsteps: int) -> int:
        acc = np.zeros((nsteps, self.N), np.uint8) if self.want_accept else None
        self._bind_user()
        st = self.lib.ko_run(C.byref(self.desc), C.byref(self.layout), self._p(self.X), self._p(self.G),
                             self._p(self.LT), self._p(self.step), self._p(self.accepted), self._p(self.proposed),
                             self._p(self.totproposed), self.t, int(nsteps), self._p(acc), self._p(self._sum),
                             self._p(self._sumsq), self._p(self.naccept), self._p(self.hist), self.hist_cols,
                             self._p(self.hist_lt), self._p(self.hist_g), self._p(self.da_epsbar), self._p(self.da_hbar),
                             self._p(self.held))
        self.t += int(nsteps)
        if acc is not None:
            self.accept = np.concatenate([self.accept, acc], axis=0)
        return st

    def run_with_batch_means(self, nsteps: int, batchlen: int):
        """Steps to nsteps, closing a batch of saved samples every `batchlen` of them (klara_desc.bm_batchlen);
        returns (mcvar_bm (N x D), nbatches) — mcvar.jl:35-41."""
        d = self.desc
        prev = np.zeros_like(self._sum); mean = np.zeros_like(self._sum); m2 = np.zeros_like(self._sum)
        nb = 0
        while self.t < nsteps:
            close_at = d.burnin + ((nb + 1) * batchlen - 1) * d.thinning + 1
            k = nsteps - self.t
            if self.t < close_at <= d.nsteps:
                k = min(k, close_at - self.t)
            assert self.run(k) == 0
            if self.t == close_at and close_at <= d.nsteps:
                view = np.ascontiguousarray(self.sum)
                self.lib.ko_bm_close(self._p(view), self._p(prev), self._p(mean), self._p(m2), view.size, nb, int(batchlen))
                nb += 1
        mcvar = batchlen * (m2 / (nb - 1)) / (nb * batchlen) if nb > 1 else np.full_like(m2, np.nan)
        return mcvar, nb

    def eval_target(self, x):
        x = _f64(x).ravel()
        lt = C.c_double(0.0)
        g = np.zeros(self.D)
        self._bind_user()
        st = self.lib.ko_eval_target(C.byref(self.desc), C.byref(self.layout), x.ctypes.data, C.byref(lt), g.ctypes.data)
        assert st == 0
        return lt.value, g


def philox_block(ctr, key):
    lib = load()
    c = (C.c_uint32 * 4)(*ctr); k = (C.c_uint32 * 2)(*key); o = (C.c_uint32 * 4)()
    lib.ko_philox_block(c, k, o)
    return list(o)


def stream_blocks(seed, chain, t, slots):
    lib = load()
    out = np.zeros((len(slots), 4), np.uint32)
    o = (C.c_uint32 * 4)()
    for i, s in enumerate(slots):
        lib.ko_stream_block(int(seed), int(chain), int(t), int(s), o)
        out[i] = list(o)
    return out


def math_op(op, x, y=None):
    lib = load()
    x = _f64(x); out = np.empty_like(x)
    y = x if y is None else _f64(y)
    lib.ko_math(int(op), x.size, x.ctypes.data, y.ctypes.data, out.ctypes.data)
    return out
