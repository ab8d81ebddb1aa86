"""CPU tests of the post-hoc chain statistics (src/stats/**) against direct formulas and AR(1) theory."""
import numpy as np
import pytest

import klara_jl_amd as K
from klara_jl_amd import stats as S


def _ar1(n, phi, seed):
    rng = np.random.default_rng(seed)
    e = rng.standard_normal(n)
    x = np.empty(n)
    x[0] = e[0] / np.sqrt(1 - phi * phi)
    for t in range(1, n):
        x[t] = phi * x[t - 1] + e[t]
    return x


def test_autocov_matches_direct_sum():
    v = np.random.default_rng(0).standard_normal(257)
    z = v - v.mean()
    direct = np.array([np.sum(z[: v.size - k] * z[k:]) / v.size for k in range(20)])
    assert np.allclose(S.autocov(v, 19), direct, rtol=1e-10, atol=1e-12)


def test_iid_and_batch_means():
    v = np.random.default_rng(1).standard_normal(1000)
    assert S.mcvar(v, "iid") == pytest.approx(v.var(ddof=1) / 1000)           # mcvar.jl:5
    bm = v.reshape(10, 100).mean(axis=1)
    assert S.mcvar(v, "bm") == pytest.approx(100 * bm.var(ddof=1) / 1000)     # mcvar.jl:35-41 (batchlen 100)
    assert S.mcvar(v, "bm", 50) == pytest.approx(50 * v.reshape(20, 50).mean(axis=1).var(ddof=1) / 1000)
    with pytest.raises(AssertionError):
        S.mcvar(v[:150], "bm")


def test_geyer_estimators_on_ar1():
    phi = 0.7
    x = _ar1(200000, phi, 2)
    tau = (1 + phi) / (1 - phi)                                               # integrated autocorrelation time
    for vt in ("imse", "ipse"):
        assert S.iact(x, vt) == pytest.approx(tau, rel=0.08)
        assert S.ess(x, vt) == pytest.approx(x.size / tau, rel=0.08)
    assert S.mcse(x, "imse") == pytest.approx(np.sqrt(x.var() * tau / x.size), rel=0.06)
    assert S.mcvar(x, "imse") <= S.mcvar(x, "ipse") + 1e-18                   # monotone sequence <= positive sequence
    iid = np.random.default_rng(3).standard_normal(50000)
    assert S.iact(iid) == pytest.approx(1.0, abs=0.1)


def test_per_dimension_helpers():
    v = np.vstack([_ar1(5000, 0.5, 4), np.random.default_rng(5).standard_normal(5000)])
    e = S.ess_chain(v)
    assert e.shape == (2,) and e[0] < e[1]
    assert np.allclose(S.mcvar_chain(v, "iid"), v.var(axis=1, ddof=1) / 5000)
    assert hasattr(K, "stats")
