"""Post-hoc chain statistics of the reference (src/stats/**) on the saved values of a chain.

These consume the path's output.  `mean`, iid variance and `acceptance` for ALL chains come from
the on-device running sums / accept masks (api.py), and every estimator below also exists on the device for every (chain, dimension)
series at once — post hoc over a stored history (klara_get_chain_mcvar / _ipse) and, for :bm / :imse / :ipse, accumulated while
sampling (klara_get_chain_bm, klara_get_chain_acov_mcvar).  This module is the NumPy restatement of the reference's estimators on one
chain's history: what the device versions are tested against.

  mcvar(v, "iid")            var(v)/length(v)                                  stats/variance/mcvar.jl:5
  mcvar(v, "bm", batchlen)   batch means, Flegal & Jones 2010                  mcvar.jl:35-41
  mcvar(v, "imse"|"ipse")    Geyer's initial monotone / positive sequence      mcvar.jl:75-105, 137-158
  mcse = sqrt(mcvar);  ess = len*iidvar/mcvar (convergence/ess.jl:3);  iact = mcvar/iidvar (convergence/iact.jl:3)
`v` is a 1-D series; the `*_chain` helpers apply an estimator to every dimension of a (D x n) NState value matrix.
"""
from __future__ import annotations

import numpy as np


def autocov(v: np.ndarray, maxlag: int) -> np.ndarray:
    """StatsBase.autocov(v, 0:maxlag) (demean=true): sum_t (v_t - m)(v_{t+k} - m) / n."""
    v = np.asarray(v, dtype=np.float64)
    n = v.size
    z = v - v.mean()
    nfft = 1 << int(np.ceil(np.log2(2 * n)))
    f = np.fft.rfft(z, nfft)
    ac = np.fft.irfft(f * np.conj(f), nfft)[: maxlag + 1]
    return ac / n


def mcvar(v, vtype: str = "imse", *args) -> float:
    v = np.asarray(v, dtype=np.float64).ravel()
    n = v.size
    if vtype == "iid":
        return float(v.var(ddof=1) / n)
    if vtype == "bm":
        batchlen = int(args[0]) if args else 100
        nbatches = n // batchlen
        assert nbatches > 1, "Choose batch size such that the number of batches is greather than one"
        nbsamples = nbatches * batchlen
        bm = v[:nbsamples].reshape(nbatches, batchlen).mean(axis=1)
        return float(batchlen * bm.var(ddof=1) / nbsamples)
    if vtype in ("imse", "ipse"):
        maxlag = int(args[0]) if args else n - 1
        k = int(np.floor((maxlag - 1) / 2))
        acv = autocov(v, maxlag)
        g = np.empty(k + 1)
        m = k + 1
        for j in range(k + 1):
            g[j] = acv[2 * j] + acv[2 * j + 1]
            if g[j] <= 0:
                m = j
                break
        if vtype == "imse" and m > 1:
            for j in range(1, m):
                if g[j] > g[j - 1]:
                    g[j] = g[j - 1]
        return float((-acv[0] + 2.0 * g[:m].sum()) / n)
    raise ValueError(f"unknown variance type {vtype!r}")


def mcse(v, vtype: str = "imse", *args) -> float:
    return float(np.sqrt(mcvar(v, vtype, *args)))


def ess(v, vtype: str = "imse", *args) -> float:
    v = np.asarray(v, dtype=np.float64).ravel()
    return float(v.size * mcvar(v, "iid") / mcvar(v, vtype, *args))


def iact(v, vtype: str = "imse", *args) -> float:
    return float(mcvar(v, vtype, *args) / mcvar(v, "iid"))


def _per_dim(fn, value: np.ndarray, *args) -> np.ndarray:
    value = np.asarray(value)
    return np.array([fn(value[i, :], *args) for i in range(value.shape[0])])


def mcvar_chain(value, vtype="imse", *args):
    """mcvar(s::VariableNState{Multivariate}, Val{vtype}) for one chain's (D x n) value matrix."""
    return _per_dim(mcvar, value, vtype, *args)


def mcse_chain(value, vtype="imse", *args):
    return _per_dim(mcse, value, vtype, *args)


def ess_chain(value, vtype="imse", *args):
    return _per_dim(ess, value, vtype, *args)


def iact_chain(value, vtype="imse", *args):
    return _per_dim(iact, value, vtype, *args)
