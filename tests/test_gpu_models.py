"""GPU tests (-m gpu) of the two data models of BASELINE cfg 4 / cfg 5 at per-GPU scale: posterior moments
against independent references (Laplace approximation for the swiss logistic regression; the published BUGS
results for the Rats growth-curve model)."""
import numpy as np
import pytest
from scipy import optimize

import cases
import klara_jl_amd as K
from klara_jl_amd import _lib as L

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("gpu_required")]


def test_swiss_logistic_mala_posterior():
    """BASELINE cfg 4 shape (32,768 chains = one GPU's share of 262,144; MALA driftstep 0.1 as in
    doc/examples/swiss/MALA/analytical.jl:30, 10000 steps in the example, 1500 here after the chains start near
    the mode).  Reference: Laplace approximation (MAP + inverse Hessian) from SciPy on the same data; the
    posterior is close to Gaussian, tolerance 0.15 posterior sd on the mean and 15 % on the sd."""
    X, y = cases.swiss_data()
    lam = 100.0

    def nlp(p):
        xp = X @ p
        return -(xp @ y - np.sum(np.logaddexp(0.0, xp)) - 0.5 * p @ p / lam)

    res = optimize.minimize(nlp, np.zeros(4), method="BFGS")
    pm = res.x
    s = 1.0 / (1.0 + np.exp(-(X @ pm)))
    hess = X.T @ (X * (s * (1 - s))[:, None]) + np.eye(4) / lam
    sd = np.sqrt(np.diag(np.linalg.inv(hess)))

    n = 32768
    x0 = pm[None, :] + 0.1 * np.random.default_rng(0).standard_normal((n, 4))
    eng = K.Engine(sampler=L.SAMPLER_MALA, target=K.LogisticTarget(X, y, lam), nchains=n, nsteps=1500, burnin=500,
                   driftstep=0.1, monitor=L.MON_SUMMARIES, steps_per_launch=50)
    eng.set_state(x0)
    eng.run(1500)
    sm, sq, na, nt, ns = eng.pooled_summaries()
    cnt = ns * n
    mean = sm / cnt
    std = np.sqrt(sq / cnt - mean * mean)
    assert 0.2 < na / nt < 0.98
    assert np.all(np.abs(mean - pm) < 0.15 * sd + 0.05 * np.abs(pm)), (mean, pm, sd)
    assert np.all(np.abs(std / sd - 1.0) < 0.3), (std, sd)
    eng.close()


def test_rats_hierarchical_hmc_posterior():
    """BASELINE cfg 5 shape (HMC L=32, per-GPU pooled AcceptanceRateMCTuner) on 16,384 chains.  Reference: the
    published WinBUGS 'Rats' results (alpha.c 242.5 (sd 2.7) at the centred age, beta.c 6.19 (0.11),
    sigma.c 6.09 (0.46)); tolerance ~0.4 posterior sd."""
    t = cases.rats_target()
    n = 16384
    x0 = t.least_squares_start()[None, :] + 0.05 * np.random.default_rng(1).standard_normal((n, t.ndims))
    eng = K.Engine(sampler=L.SAMPLER_HMC, target=t, nchains=n, nsteps=700, burnin=400, leapstep=0.02, nleaps=32,
                   tuner=L.TUNER_ACCEPT_RATE, tuner_mode=L.TUNE_POOLED, targetrate=0.65, period=50,
                   monitor=L.MON_SUMMARIES, steps_per_launch=10)
    eng.set_state(x0)
    eng.run(700)
    sm, sq, na, nt, ns = eng.pooled_summaries()
    mean = sm / (ns * n)
    r = 30
    step = eng.tune()[0][0]
    assert 0.4 < na / nt < 0.95, (na / nt, step)
    assert abs(mean[2 * r] - 242.6) < 1.2, mean[2 * r]              # alpha_c
    assert abs(mean[2 * r + 1] - 6.186) < 0.06, mean[2 * r + 1]     # beta_c
    assert abs(mean[2 * r + 2] - np.log(6.07)) < 0.06, mean[2 * r + 2]   # log sigma_c
    # the unit-level slopes average to beta_c
    assert abs(mean[1:2 * r:2].mean() - mean[2 * r + 1]) < 0.05
    eng.close()
