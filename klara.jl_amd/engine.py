"""Thin object wrapper over the C ABI (one Engine = one klara_handle = N chains on one GPU).

Host buffers are NumPy arrays; nothing here computes a transition — every call forwards to
libklara_hip.so (see include/klara_hip.h for the reference lines each entry point replaces).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import _lib as L


def _f64(a, shape=None) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and a.shape != tuple(shape):
        raise ValueError(f"expected shape {tuple(shape)}, got {a.shape}")
    return a


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


# ------------------------------------------------------------------ target families
@dataclass
class GaussDiagTarget:
    """lt = c - sum_i w_i (x_i - mu_i)^2 (KLARA_TARGET_GAUSS_DIAG).

    README.md:23 `plogtarget(z) = -dot(z, z)` is `GaussDiagTarget.negdot(D)`;
    MvNormal(mu, sigma) closures of test/BasicContMuvParameter.jl:39-56,88-100 are `mvnormal(mu, sigma)`.
    """
    ndims: int
    w: Optional[np.ndarray] = None
    mu: Optional[np.ndarray] = None
    const: float = 0.0
    kind = L.TARGET_GAUSS_DIAG

    @classmethod
    def negdot(cls, ndims: int) -> "GaussDiagTarget":
        return cls(int(ndims))

    @classmethod
    def mvnormal(cls, mu, sigma) -> "GaussDiagTarget":
        mu = _f64(mu).ravel()
        sigma = np.broadcast_to(_f64(sigma).ravel(), mu.shape).astype(np.float64)
        d = mu.size
        const = -0.5 * d * np.log(2.0 * np.pi) - float(np.sum(np.log(sigma)))
        return cls(d, w=1.0 / (2.0 * sigma * sigma), mu=mu.copy(), const=float(const))


@dataclass
class GaussDenseTarget:
    """lt = c - 1/2 (x-mu)' P (x-mu) with a dense D x D precision matrix (KLARA_TARGET_GAUSS_DENSE, FP64 MFMA); mu = None is 0."""
    precision: np.ndarray
    const: float = 0.0
    mu: Optional[np.ndarray] = None
    kind = L.TARGET_GAUSS_DENSE

    def __post_init__(self):
        self.precision = _f64(self.precision)
        if self.precision.ndim != 2 or self.precision.shape[0] != self.precision.shape[1]:
            raise ValueError("precision must be square")
        if self.mu is not None:
            self.mu = _f64(self.mu)
            if self.mu.shape != (self.precision.shape[0],):
                raise ValueError("mu must have one entry per dimension")

    @property
    def ndims(self) -> int:
        return int(self.precision.shape[0])

    @classmethod
    def compound_symmetric(cls, ndims: int, rho: float = 0.5) -> "GaussDenseTarget":
        """Sigma = (1-rho) I + rho 11' ; closed-form inverse (SURVEY §8(d) cfg 3)."""
        d = int(ndims)
        prec = (np.eye(d) - (rho / (1.0 - rho + d * rho)) * np.ones((d, d))) / (1.0 - rho)
        return cls(prec)


@dataclass
class LogisticTarget:
    """Bayesian logistic regression, N(0, lambda I) prior (doc/examples/swiss/MALA/analytical.jl:11-18)."""
    X: np.ndarray
    y: np.ndarray
    lam: float = 100.0
    kind = L.TARGET_LOGISTIC

    def __post_init__(self):
        self.X = _f64(self.X)
        self.y = _f64(self.y).ravel()
        if self.X.ndim != 2 or self.X.shape[0] != self.y.size:
            raise ValueError("X must be (ndata, D) and y (ndata,)")

    @property
    def ndims(self) -> int:
        return int(self.X.shape[1])


@dataclass
class CustomTarget:
    """User-defined target: the device form of `BasicContMuvParameter(:p, logtarget=f, gradlogtarget=g)`
    (BasicContMuvParameter.jl:174-201,264-279).  `source` is C text defining

        KLARA_USER_FN double klara_user_logtarget(const double* x, int D, const double* data, long long ndata);
        KLARA_USER_FN void   klara_user_gradlogtarget(const double* x, int D, const double* data, long long ndata, double* g);

    (the gradient only for MALA / HMC); it is compiled for gfx950 when the job is created (include/klara_hip.h,
    KLARA_TARGET_CUSTOM).  `data` is an optional read-only block of doubles handed to both functions."""
    ndims: int
    source: str
    data: Optional[np.ndarray] = None
    kind = L.TARGET_CUSTOM

    def __post_init__(self):
        self.ndims = int(self.ndims)
        if self.data is not None:
            self.data = _f64(self.data).ravel()

    @classmethod
    def likelihood_prior(cls, ndims: int, loglikelihood: str, logprior: str, gradloglikelihood: str = "", gradlogprior: str = "",
                         data=None) -> "CustomTarget":
        """The likelihood + prior form of `BasicContMuvParameter(:p, loglikelihood=..., logprior=..., gradloglikelihood=...,
        gradlogprior=...)` (BasicContMuvParameter.jl:174-201): each argument is C text defining klara_user_loglikelihood /
        klara_user_logprior / klara_user_gradloglikelihood / klara_user_gradlogprior; the library composes
        logtarget = loglikelihood + logprior and gradlogtarget = gradloglikelihood + gradlogprior (klara_custom_compose.h)."""
        src = "#define KLARA_USER_LIKELIHOOD_PRIOR 1\n" + "\n".join(t for t in (loglikelihood, logprior, gradloglikelihood, gradlogprior) if t)
        return cls(ndims, src, data)

    @classmethod
    def pairwise(cls, ndims: int, pair_source: str, data=None) -> "CustomTarget":
        """Closures of a target that is a SUM OF TERMS OF ONE OR TWO NEIGHBOURING COORDINATES, one element pair at a time:
        `pair_source` is C text defining

            KLARA_USER_FN double klara_user_pair(double x0, double x1, int pair, int D, const double* data, long long ndata,
                                                 double* g0, double* g1);

        with logtarget(x) = sum over pairs P of klara_user_pair(x[2P], x[2P+1], P, ...) and (*g0, *g1) the pair's two partial
        derivatives (for the half pair of an odd D, x1 is 0 and *g1 is ignored).  Such a job runs on the few-lanes-per-chain
        kernels of the diagonal Gaussian (layout kind 3: 8 / 16 / 32 / 64 lanes per chain, 17 <= D <= 1024; MH, MALA, HMC with every tuner
        and monitor) instead of one chain per lane — the form for large D (include/klara_hip.h, KLARA_USER_PAIR_TARGET).  Below 17 dimensions, and
        the library sums the pairs' terms itself and runs the whole-vector form (D <= 1024); the slice sampler runs on them too (round 6: a probe compares the pair's own term)."""
        return cls(ndims, "#define KLARA_USER_PAIR_TARGET 1\n" + pair_source, data)

    @property
    def has_parts(self) -> bool:
        return "KLARA_USER_LIKELIHOOD_PRIOR" in self.source

    @property
    def is_pairwise(self) -> bool:
        return "KLARA_USER_PAIR_TARGET" in self.source

    def check(self, sampler: int) -> None:
        """Compile only (no GPU needed); raises KlaraError with the compiler's log on failure."""
        L.check(L.load().klara_check_custom_target(self.source.encode(), int(sampler), self.ndims), "klara_check_custom_target")


@dataclass
class HierNormalTarget:
    """Hierarchical normal growth-curve model (BUGS "Rats"; BASELINE cfg 5, data/rats/*.csv).

    theta = (alpha_1, beta_1, ..., alpha_R, beta_R, alpha_c, beta_c, log sigma_c, log sigma_alpha, log sigma_beta),
    D = 2R + 5.  Builder-defined: the reference ships only the data (doc/examples/rats/Gibbs.jl is a stub).
    """
    Y: np.ndarray                 # (R, T) observations
    xc: np.ndarray                # (T,) centred covariate (age - 22 for the rats)
    prior_prec: float = 1e-4      # alpha_c, beta_c ~ N(0, 1/prior_prec)
    gamma_a: float = 1e-3         # precisions ~ Gamma(a, b)
    gamma_b: float = 1e-3
    kind = L.TARGET_HIER_NORMAL

    def __post_init__(self):
        self.Y = _f64(self.Y)
        self.xc = _f64(self.xc).ravel()
        if self.Y.ndim != 2 or self.Y.shape[1] != self.xc.size:
            raise ValueError("Y must be (R, T) and xc (T,)")

    @property
    def ndims(self) -> int:
        return 2 * int(self.Y.shape[0]) + 5

    def least_squares_start(self) -> np.ndarray:
        """Per-unit OLS fit -> a reasonable initial theta (SURVEY §8(d) cfg 5: 'least-squares fit + jitter')."""
        r, t = self.Y.shape
        sxx = float(self.xc @ self.xc)
        xbar = float(self.xc.mean())
        beta = ((self.Y - self.Y.mean(axis=1, keepdims=True)) @ (self.xc - xbar)) / float((self.xc - xbar) @ (self.xc - xbar))
        alpha = self.Y.mean(axis=1) - beta * xbar
        resid = self.Y - alpha[:, None] - beta[:, None] * self.xc[None, :]
        th = np.empty(2 * r + 5)
        th[0:2 * r:2] = alpha
        th[1:2 * r:2] = beta
        th[2 * r] = alpha.mean(); th[2 * r + 1] = beta.mean()
        th[2 * r + 2] = np.log(resid.std(ddof=2) + 1e-12)
        th[2 * r + 3] = np.log(alpha.std(ddof=1) + 1e-12)
        th[2 * r + 4] = np.log(beta.std(ddof=1) + 1e-12)
        del sxx, t
        return th


# ------------------------------------------------------------------ engine
class Engine:
    def __init__(self, *, sampler: int, target, nchains: int, nsteps: int, burnin: int = 0, thinning: int = 1,
                 mh_sigma=None, driftstep: float = 1.0, leapstep: float = 0.1, nleaps: int = 10,
                 slice_widths=None, slice_stepout: bool = True,
                 tuner: int = L.TUNER_VANILLA, tuner_mode: int = L.TUNE_PER_CHAIN, targetrate: float = 0.0,
                 score_k: float = 7.0, tuner_score: int = 0, period: int = 100, verbose: bool = False,
                 da_nadapt: int = 0, da_eps0bar: float = 1.0, da_h0bar: float = 0.0, da_gamma: float = 0.05,
                 da_t0: int = 10, da_kappa: float = 0.75,
                 seed: int = 20260927, chain_offset: int = 0, device: int = 0, monitor: int = 0,
                 steps_per_launch: int = 0, stream: int = 0, nstreams: int = 0, bm_batchlen: int = 0, hist_ring_cols: int = 0,
                 acov_maxlag: int = 0, sparse_moves: int = 0):
        self._lib = L.load()
        self.target = target
        self.ndims = int(target.ndims)
        self.nchains = int(nchains)
        self.nsteps, self.burnin, self.thinning = int(nsteps), int(burnin), int(thinning)
        self.sampler = int(sampler)
        self.monitor = int(monitor)
        d = L.KlaraDesc()
        d.struct_size = C.sizeof(L.KlaraDesc)
        d.abi_version = getattr(L.load(), "_klara_abi_override", L.KLARA_ABI_VERSION)   # (an older build under KLARA_ALLOW_ABI_MISMATCH=1: its own version)
        d.sampler, d.target, d.tuner, d.tuner_mode = int(sampler), int(target.kind), int(tuner), int(tuner_mode)
        d.nchains, d.chain_offset, d.ndims, d.device = self.nchains, int(chain_offset), self.ndims, int(device)
        keep = []  # keep host arrays alive until klara_create returns
        if mh_sigma is not None:
            a = _f64(np.broadcast_to(_f64(mh_sigma).ravel(), (self.ndims,))); keep.append(a); d.mh_sigma = _ptr(a)
        if slice_widths is not None:
            a = _f64(np.broadcast_to(_f64(slice_widths).ravel(), (self.ndims,))); keep.append(a); d.slice_widths = _ptr(a)
        d.driftstep, d.leapstep, d.nleaps, d.slice_stepout = float(driftstep), float(leapstep), int(nleaps), int(bool(slice_stepout))
        d.targetrate, d.score_k, d.period, d.verbose = float(targetrate), float(score_k), int(period), int(bool(verbose))
        d.nsteps, d.burnin, d.thinning = self.nsteps, self.burnin, self.thinning
        d.da_nadapt, d.da_eps0bar, d.da_h0bar = int(da_nadapt), float(da_eps0bar), float(da_h0bar)
        d.da_gamma, d.da_kappa, d.da_t0 = float(da_gamma), float(da_kappa), int(da_t0)
        d.tuner_score = int(tuner_score)
        if isinstance(target, GaussDiagTarget):
            if target.w is not None:
                a = _f64(target.w, (self.ndims,)); keep.append(a); d.gauss_w = _ptr(a)
            if target.mu is not None:
                a = _f64(target.mu, (self.ndims,)); keep.append(a); d.gauss_mu = _ptr(a)
            d.gauss_const = float(target.const)
        elif isinstance(target, GaussDenseTarget):
            a = _f64(target.precision); keep.append(a); d.gauss_prec = _ptr(a)
            if target.mu is not None:
                a = _f64(target.mu, (self.ndims,)); keep.append(a); d.gauss_mu = _ptr(a)
            d.gauss_const = float(target.const)
        elif isinstance(target, LogisticTarget):
            a = _f64(target.X); keep.append(a); d.logit_X = _ptr(a)
            b = _f64(target.y); keep.append(b); d.logit_y = _ptr(b)
            d.logit_ndata, d.logit_lambda = int(target.X.shape[0]), float(target.lam)
        elif isinstance(target, HierNormalTarget):
            a = _f64(target.Y); keep.append(a); d.hier_Y = _ptr(a)
            b = _f64(target.xc); keep.append(b); d.hier_xc = _ptr(b)
            d.hier_nunits, d.hier_ntimes = int(target.Y.shape[0]), int(target.Y.shape[1])
            d.hier_prior_prec, d.hier_gamma_a, d.hier_gamma_b = float(target.prior_prec), float(target.gamma_a), float(target.gamma_b)
        elif isinstance(target, CustomTarget):
            src = target.source.encode(); keep.append(src); d.custom_src = src
            if target.data is not None and target.data.size:
                a = _f64(target.data); keep.append(a); d.custom_data = _ptr(a); d.custom_ndata = int(a.size)
        else:
            raise TypeError(f"unknown target family {type(target).__name__}")
        d.seed, d.monitor, d.steps_per_launch = int(seed), self.monitor, int(steps_per_launch)
        d.nstreams = int(nstreams)
        d.bm_batchlen = int(bm_batchlen)
        d.hist_ring_cols, d.acov_maxlag, d.sparse_moves = int(hist_ring_cols), int(acov_maxlag), int(sparse_moves)
        d.stream = C.c_void_p(int(stream)) if stream else None
        self._h = C.c_void_p()
        L.check(self._lib.klara_create(C.byref(d), C.byref(self._h)), "klara_create")
        del keep

    # -- lifetime
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            h, self._h = self._h, C.c_void_p()
            L.check(self._lib.klara_destroy(h), "klara_destroy")     # (KLARA_DEBUG_CANARY=1: fails when a kernel wrote outside an array)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- state
    def set_state(self, x):
        x = _f64(x, (self.nchains, self.ndims))
        L.check(self._lib.klara_set_state(self._h, x.ctypes.data), "klara_set_state")

    def init_state_normal(self):
        L.check(self._lib.klara_init_state_normal(self._h), "klara_init_state_normal")

    def reset(self, x=None):
        if x is None:
            L.check(self._lib.klara_reset(self._h, None), "klara_reset")
        else:
            x = _f64(x, (self.nchains, self.ndims))
            L.check(self._lib.klara_reset(self._h, x.ctypes.data), "klara_reset")

    def stream_key(self):
        """(Philox key the job currently draws from, klara_reset calls so far) — include/klara_hip.h klara_reset."""
        k, e = C.c_uint64(0), C.c_uint64(0)
        L.check(self._lib.klara_stream_key(self._h, C.byref(k), C.byref(e)), "klara_stream_key")
        return int(k.value), int(e.value)

    def run(self, nsteps: int):
        L.check(self._lib.klara_run(self._h, int(nsteps)), "klara_run")

    def run_async(self, nsteps: int):
        L.check(self._lib.klara_run_async(self._h, int(nsteps)), "klara_run_async")

    def synchronize(self):
        L.check(self._lib.klara_synchronize(self._h), "klara_synchronize")

    # -- read-back
    def state(self):
        x = np.empty((self.nchains, self.ndims)); lt = np.empty(self.nchains); g = np.empty((self.nchains, self.ndims))
        L.check(self._lib.klara_get_state(self._h, x.ctypes.data, lt.ctypes.data, g.ctypes.data), "klara_get_state")
        return x, lt, g

    def accept_mask(self) -> np.ndarray:
        n = C.c_int64(0)
        L.check(self._lib.klara_get_accept_mask(self._h, None, 0, C.byref(n)), "klara_get_accept_mask")
        m = np.empty((n.value, self.nchains), dtype=np.uint8)
        L.check(self._lib.klara_get_accept_mask(self._h, m.ctypes.data, n.value, C.byref(n)), "klara_get_accept_mask")
        return m

    def accept_rows(self, first_step: int, nsteps: int) -> np.ndarray:
        """accept diagnostics of the transitions [first_step, first_step + nsteps) only: (nsteps, nchains) bytes"""
        m = np.empty((int(nsteps), self.nchains), dtype=np.uint8)
        L.check(self._lib.klara_get_accept_rows(self._h, int(first_step), int(nsteps), m.ctypes.data), "klara_get_accept_rows")
        return m

    def accept_counts(self):
        a = np.empty(self.nchains, dtype=np.uint64); n = C.c_uint64(0)
        L.check(self._lib.klara_get_accept_counts(self._h, a.ctypes.data, C.byref(n)), "klara_get_accept_counts")
        return a, int(n.value)

    def chain_sums(self):
        s = np.empty((self.nchains, self.ndims)); q = np.empty((self.nchains, self.ndims)); n = C.c_int64(0)
        L.check(self._lib.klara_get_chain_sums(self._h, s.ctypes.data, q.ctypes.data, C.byref(n)), "klara_get_chain_sums")
        return s, q, int(n.value)

    def pooled_summaries(self, with_sums: bool = True):
        s = np.empty(self.ndims) if with_sums else None
        q = np.empty(self.ndims) if with_sums else None
        na, nt, ns = C.c_uint64(0), C.c_uint64(0), C.c_int64(0)
        L.check(self._lib.klara_get_pooled_summaries(
            self._h, s.ctypes.data if with_sums else None, q.ctypes.data if with_sums else None,
            C.byref(na), C.byref(nt), C.byref(ns)), "klara_get_pooled_summaries")
        return s, q, int(na.value), int(nt.value), int(ns.value)

    def pooled_moments(self, comm=None):
        """(mean[D], M2[D], nsamples, naccept, ntransitions, nchains) over this handle's chains — or, with a klara_comm handle,
        over every rank's — formed on the device without the cancellation of sumsq/n - mean^2 (klara_gather_moments)."""
        mean = np.empty(self.ndims); m2 = np.empty(self.ndims)
        ns, na, nt, nc = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        L.check(self._lib.klara_gather_moments(self._h, comm, mean.ctypes.data, m2.ctypes.data, C.byref(ns), C.byref(na), C.byref(nt),
                                               C.byref(nc)), "klara_gather_moments")
        return mean, m2, int(ns.value), int(na.value), int(nt.value), int(nc.value)

    def chain(self, local_chain: int) -> np.ndarray:
        """One chain's saved values in Klara's NState layout: (ndims, nsaved), column-major."""
        n = C.c_int64(0)
        L.check(self._lib.klara_get_chain(self._h, int(local_chain), None, 0, C.byref(n)), "klara_get_chain")
        v = np.empty((self.ndims, n.value), order="F")
        L.check(self._lib.klara_get_chain(self._h, int(local_chain), v.ctypes.data, n.value, C.byref(n)), "klara_get_chain")
        return v

    def chain_fields(self, local_chain: int, logtarget: bool = True, gradlogtarget: bool = False):
        """logtarget (n,) and/or gradlogtarget (ndims, n) of one chain over the saved steps (NState fields)."""
        n = C.c_int64(0)
        L.check(self._lib.klara_get_chain_fields(self._h, int(local_chain), None, None, 0, C.byref(n)), "klara_get_chain_fields")
        lt = np.empty(n.value) if logtarget else None
        g = np.empty((self.ndims, n.value), order="F") if gradlogtarget else None
        L.check(self._lib.klara_get_chain_fields(self._h, int(local_chain), lt.ctypes.data if logtarget else None,
                                                 g.ctypes.data if gradlogtarget else None, n.value, C.byref(n)),
                "klara_get_chain_fields")
        return lt, g

    def chain_likelihood_prior(self, local_chain: int):
        """(loglikelihood (n,), logprior (n,)) of one chain over the saved steps — a likelihood + prior user target monitored with
        MON_HIST_LLLP (:monitor => [:loglikelihood, :logprior])."""
        n = C.c_int64(0)
        L.check(self._lib.klara_get_chain_likelihood_prior(self._h, int(local_chain), None, None, 0, C.byref(n)), "klara_get_chain_likelihood_prior")
        ll = np.empty(n.value); lp = np.empty(n.value)
        L.check(self._lib.klara_get_chain_likelihood_prior(self._h, int(local_chain), ll.ctypes.data, lp.ctypes.data, n.value, C.byref(n)),
                "klara_get_chain_likelihood_prior")
        return ll, lp

    def chain_mcvar(self, batchlen: int = 100, maxlag: int = 0, want=("iid", "bm", "imse")):
        """(mcvar_iid, mcvar_bm, mcvar_imse), each (nchains, ndims) or None when not in `want`, from the on-device history
        (mcvar.jl); an estimator that is not asked for is not computed (the IMSE pass costs O(n x stopping lag) per series)."""
        out = [np.empty((self.nchains, self.ndims)) if k in want else None for k in ("iid", "bm", "imse")]
        L.check(self._lib.klara_get_chain_mcvar(self._h, int(batchlen), int(maxlag), *[None if a is None else a.ctypes.data for a in out]),
                "klara_get_chain_mcvar")
        return tuple(out)

    def chain_acov_mcvar(self, want=("imse", "ipse")):
        """(mcvar_imse, mcvar_ipse, nsamples), each (nchains, ndims) or None: Geyer's estimators with maxlag = acov_maxlag from the
        autocovariances accumulated while sampling — no stored history (mcvar.jl:75-105, 137-158)."""
        out = [np.empty((self.nchains, self.ndims)) if k in want else None for k in ("imse", "ipse")]
        n = C.c_int64(0)
        L.check(self._lib.klara_get_chain_acov_mcvar(self._h, *[None if a is None else a.ctypes.data for a in out], C.byref(n)),
                "klara_get_chain_acov_mcvar")
        return out[0], out[1], int(n.value)

    def chain_mcvar_ipse(self, maxlag: int = 0) -> np.ndarray:
        """mcvar(:ipse, maxlag) of every (chain, dimension) series over the stored history (mcvar.jl:137-158)."""
        out = np.empty((self.nchains, self.ndims))
        L.check(self._lib.klara_get_chain_mcvar_ipse(self._h, int(maxlag), out.ctypes.data), "klara_get_chain_mcvar_ipse")
        return out

    def saved_steps(self) -> int:
        n = C.c_int64(0)
        L.check(self._lib.klara_saved_steps(self._h, C.byref(n)), "klara_saved_steps")
        return int(n.value)

    def chain_bm(self):
        """(mcvar_bm (nchains, ndims), nbatches): streaming batch means (bm_batchlen > 0), no stored history (mcvar.jl:35-41)."""
        out = np.empty((self.nchains, self.ndims)); nb = C.c_int64(0)
        L.check(self._lib.klara_get_chain_bm(self._h, out.ctypes.data, C.byref(nb)), "klara_get_chain_bm")
        return out, int(nb.value)

    def tune(self):
        step = np.empty(self.nchains); a = np.empty(self.nchains, dtype=np.int64)
        p = np.empty(self.nchains, dtype=np.int64); t = np.empty(self.nchains, dtype=np.int64)
        L.check(self._lib.klara_get_tune(self._h, step.ctypes.data, a.ctypes.data, p.ctypes.data, t.ctypes.data), "klara_get_tune")
        return step, a, p, t

    def dual_averaging(self):
        eb = np.empty(self.nchains); hb = np.empty(self.nchains)
        L.check(self._lib.klara_get_dual_averaging(self._h, eb.ctypes.data, hb.ctypes.data), "klara_get_dual_averaging")
        return eb, hb

    def last_run_ms(self):
        ms, n = C.c_double(0.0), C.c_int64(0)
        L.check(self._lib.klara_last_run_ms(self._h, C.byref(ms), C.byref(n)), "klara_last_run_ms")
        return float(ms.value), int(n.value)

    def layout(self):
        k, g, e = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        L.check(self._lib.klara_get_layout(self._h, C.byref(k), C.byref(g), C.byref(e)), "klara_get_layout")
        return int(k.value), int(g.value), int(e.value)

    def launch_modes(self):
        """(counts[3], last_mode[4], last_accepted[4]) — how the launches were issued (klara_get_launch_modes): 4-lane kernel alone,
        8-lane kernel alone, device-decided pair; the per-partition values may lag the device (never synchronises)."""
        cnt = np.zeros(3, dtype=np.int64); lm = np.zeros(4, dtype=np.int32); la = np.zeros(4, dtype=np.int64)
        L.check(self._lib.klara_get_launch_modes(self._h, cnt.ctypes.data, lm.ctypes.data, la.ctypes.data), "klara_get_launch_modes")
        return cnt, lm, la

    def kernel_attributes(self, which: int = 0, nsteps: int = 32):
        """(vgprs, scratch bytes, static LDS bytes) of the transition kernel a launch of `nsteps` transitions runs, from the loaded
        code object (klara_get_kernel_attributes); which = 1: the 8-lane sibling of jobs two kernel families can run."""
        v, s_, l = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        L.check(self._lib.klara_get_kernel_attributes(self._h, int(which), int(nsteps), C.byref(v), C.byref(s_), C.byref(l)), "klara_get_kernel_attributes")
        return int(v.value), int(s_.value), int(l.value)

    def shader_clock_mhz(self) -> float:
        """Shader clock during the last launch of a pair-transposed kernel (klara_get_shader_clock); 0.0 when unknown."""
        if not hasattr(self._lib, "klara_get_shader_clock"):
            return 0.0
        v = C.c_double(0.0)
        L.check(self._lib.klara_get_shader_clock(self._h, C.byref(v)), "klara_get_shader_clock")
        return float(v.value)

    def device_ptrs(self):
        x, lt, g = C.c_void_p(), C.c_void_p(), C.c_void_p()
        L.check(self._lib.klara_device_ptrs(self._h, C.byref(x), C.byref(lt), C.byref(g)), "klara_device_ptrs")
        return x.value, lt.value, g.value
