"""The C oracle against the independent numpy restatement (tests/numpy_mirror.py): same accept decisions at every transition, states /
log-targets / tuned steps / chain means equal to rounding.  The two share no code — not the Philox generator, not the uniform and
normal transforms (libm here, the library's table-driven functions there), not the sampler or target arithmetic — so agreement pins the
oracle's transition logic (and, through the accept masks, its random-number contract) from a second reading of the Julia sources."""
import numpy as np
import pytest

import cases
import numpy_mirror as M
import oracle_ffi as O
from klara_jl_amd import _lib as L

SEED, OFFSET, NCHAINS = 20260927, 5, 3


def _compare(job, chains, nsteps, burnin, thinning, check_grad):
    assert job.run(nsteps) == 0
    for c in chains:
        c.run(nsteps)
    nsaved = (nsteps - burnin - 1) // thinning + 1 if nsteps > burnin else 0
    for k, c in enumerate(chains):
        assert np.array_equal(np.asarray(job.accept[:, k], bool), np.array(c.accepts)), ("accept decisions differ", k)
        assert np.allclose(job.X[k], c.x, rtol=1e-9, atol=1e-11), k
        assert job.LT[k] == pytest.approx(c.lt, rel=1e-9, abs=1e-9)
        if check_grad:
            assert np.allclose(job.G[k], c.g, rtol=1e-8, atol=1e-9)
        assert job.step[k] == pytest.approx(c.step, rel=1e-9, nan_ok=True)      # (dual averaging feeds a = min(1, exp(dH)) back into the step)
        assert (int(job.accepted[k]), int(job.proposed[k]), int(job.totproposed[k])) == (c.accepted, c.proposed, c.totproposed)
        assert len(c.saved) == nsaved
        if nsaved:
            assert np.allclose(job.sum[k] / nsaved, np.mean(c.saved, axis=0), rtol=1e-9, atol=1e-11)      # mean(chain), stats/mean.jl:7-11


def test_philox_and_uniforms_agree_with_the_library():
    for (seed, chain, t, slot) in ((SEED, 0, 0, 0), (1, (1 << 33) + 7, 12345, 49), (2 ** 63 + 11, 65535, M.INIT_T, 3), (7, 3, 9, (4 << 14) | 2)):
        blk = O.stream_blocks(seed, chain, t, [slot])[0]
        assert tuple(int(v) for v in blk) == M.stream_block(seed, chain, t, slot)
        assert O.load().ko_u52(int(blk[0]), int(blk[1])) == M.u52(int(blk[0]), int(blk[1]))
        assert O.load().ko_u44(int(blk[2]), int(blk[3])) == M.u44(int(blk[2]), int(blk[3]))


def test_mh_readme_job_with_verbose_counters():
    lt, grad = M.diag_target(np.ones(2), np.zeros(2), 0.0)
    kw = dict(nsteps=300, burnin=50, thinning=1)
    job = O.OracleJob(sampler=L.SAMPLER_MH, target_kind=L.TARGET_GAUSS_DIAG, nchains=NCHAINS, ndims=2, mh_sigma=[1.0, 1.0], verbose=True, period=7,
                      seed=SEED, chain_offset=OFFSET, **kw)
    x0 = np.tile([5.1, -0.9], (NCHAINS, 1))
    assert job.set_state(x0) == 0
    chains = [M.Chain("mh", lt, grad, x0[k], SEED, OFFSET + k, sigma=np.ones(2), verbose=True, period=7, **kw) for k in range(NCHAINS)]
    _compare(job, chains, 300, 50, 1, False)


def test_mala_with_the_acceptance_rate_tuner_and_thinning():
    d = 5
    lt, grad = M.diag_target(np.ones(d), np.zeros(d), 0.0)
    kw = dict(nsteps=120, burnin=40, thinning=3)
    job = O.OracleJob(sampler=L.SAMPLER_MALA, target_kind=L.TARGET_GAUSS_DIAG, nchains=NCHAINS, ndims=d, driftstep=0.4, tuner=L.TUNER_ACCEPT_RATE,
                      targetrate=0.574, period=5, seed=SEED, chain_offset=OFFSET, **kw)
    assert job.init_state_normal() == 0
    x0 = np.array([M.init_state_normal(SEED, OFFSET + k, d) for k in range(NCHAINS)])
    assert np.allclose(job.X, x0, rtol=1e-12, atol=1e-14)                  # x0 ~ N(0, I) from the init stream
    assert job.set_state(x0) == 0
    chains = [M.Chain("mala", lt, grad, x0[k], SEED, OFFSET + k, driftstep=0.4, tuner="rate", targetrate=0.574, period=5, **kw) for k in range(NCHAINS)]
    _compare(job, chains, 120, 40, 3, True)


def test_mala_vanilla_verbose_counts_without_tuning():
    d = 4
    lt, grad = M.diag_target(np.ones(d), np.zeros(d), 0.0)
    kw = dict(nsteps=100, burnin=60, thinning=1)
    job = O.OracleJob(sampler=L.SAMPLER_MALA, target_kind=L.TARGET_GAUSS_DIAG, nchains=NCHAINS, ndims=d, driftstep=0.6, verbose=True, period=8,
                      seed=SEED, chain_offset=OFFSET, **kw)
    x0 = np.random.default_rng(12).standard_normal((NCHAINS, d))
    assert job.set_state(x0) == 0
    chains = [M.Chain("mala", lt, grad, x0[k], SEED, OFFSET + k, driftstep=0.6, verbose=True, period=8, **kw) for k in range(NCHAINS)]
    _compare(job, chains, 100, 60, 1, True)
    assert all(c.step == 0.6 for c in chains)                     # VanillaMCTuner: counters and burn-in reports only


def test_hmc_on_the_dense_target_with_a_mean_tuned():
    d = 6
    rng = np.random.default_rng(3)
    a = rng.standard_normal((d, d)); P = a @ a.T / d + np.eye(d); mu = rng.standard_normal(d)
    lt, grad = M.dense_target(P, mu, 0.25)
    kw = dict(nsteps=90, burnin=30, thinning=1)
    job = O.OracleJob(sampler=L.SAMPLER_HMC, target_kind=L.TARGET_GAUSS_DENSE, nchains=NCHAINS, ndims=d, leapstep=0.25, nleaps=4, gauss_prec=P, gauss_mu=mu,
                      gauss_const=0.25, tuner=L.TUNER_ACCEPT_RATE, targetrate=0.65, period=10, seed=SEED, chain_offset=OFFSET, **kw)
    x0 = mu[None, :] + rng.standard_normal((NCHAINS, d))
    assert job.set_state(x0) == 0
    chains = [M.Chain("hmc", lt, grad, x0[k], SEED, OFFSET + k, leapstep=0.25, nleaps=4, tuner="rate", targetrate=0.65, period=10, **kw) for k in range(NCHAINS)]
    _compare(job, chains, 90, 30, 1, True)


def test_hmc_on_a_diagonal_mvnormal_vanilla():
    d = 7
    t = cases.K.GaussDiagTarget.mvnormal(np.linspace(-1, 1, d), np.linspace(0.6, 1.6, d))
    lt, grad = M.diag_target(t.w, t.mu, t.const)
    kw = dict(nsteps=80, burnin=0, thinning=2)
    job = O.OracleJob(sampler=L.SAMPLER_HMC, target_kind=L.TARGET_GAUSS_DIAG, nchains=NCHAINS, ndims=d, leapstep=0.3, nleaps=5, gauss_w=t.w, gauss_mu=t.mu,
                      gauss_const=t.const, seed=SEED, chain_offset=OFFSET, **kw)
    x0 = np.random.default_rng(5).standard_normal((NCHAINS, d))
    assert job.set_state(x0) == 0
    chains = [M.Chain("hmc", lt, grad, x0[k], SEED, OFFSET + k, leapstep=0.3, nleaps=5, **kw) for k in range(NCHAINS)]
    _compare(job, chains, 80, 0, 2, True)


@pytest.mark.parametrize("verbose", [False, True])
def test_hmc_with_the_dual_averaging_tuner(verbose):
    d = 5
    t = cases.K.GaussDiagTarget.mvnormal(np.linspace(-1, 1, d), np.linspace(0.6, 1.6, d))
    lt, grad = M.diag_target(t.w, t.mu, t.const)
    kw = dict(nsteps=70, burnin=40, thinning=1)
    job = O.OracleJob(sampler=L.SAMPLER_HMC, target_kind=L.TARGET_GAUSS_DIAG, nchains=NCHAINS, ndims=d, leapstep=0.25, nleaps=5, gauss_w=t.w, gauss_mu=t.mu,
                      gauss_const=t.const, tuner=L.TUNER_DUAL_AVERAGING, targetrate=0.65, da_nadapt=40, verbose=verbose, period=10,
                      seed=SEED, chain_offset=OFFSET, **kw)
    x0 = np.random.default_rng(9).standard_normal((NCHAINS, d))
    assert job.set_state(x0) == 0
    chains = [M.Chain("hmc", lt, grad, x0[k], SEED, OFFSET + k, leapstep=0.25, nleaps=5, tuner="da", targetrate=0.65, nadapt=40, verbose=verbose,
                      period=10, **kw) for k in range(NCHAINS)]
    _compare(job, chains, 70, 40, 1, True)
    for k, c in enumerate(chains):
        assert job.da_epsbar[k] == pytest.approx(c.epsbar, rel=1e-10) and job.da_hbar[k] == pytest.approx(c.hbar, rel=1e-9, abs=1e-12)


@pytest.mark.parametrize("stepout", [True, False])
def test_slice_sampler(stepout):
    d = 3
    lt, grad = M.diag_target(np.ones(d), np.zeros(d), 0.0)
    kw = dict(nsteps=40, burnin=5, thinning=1)
    w = np.array([1.0, 2.5, 0.6])
    job = O.OracleJob(sampler=L.SAMPLER_SLICE, target_kind=L.TARGET_GAUSS_DIAG, nchains=NCHAINS, ndims=d, slice_widths=w, slice_stepout=stepout,
                      seed=SEED, chain_offset=OFFSET, **kw)
    x0 = np.random.default_rng(6).standard_normal((NCHAINS, d))
    assert job.set_state(x0) == 0
    chains = [M.Chain("slice", lt, grad, x0[k], SEED, OFFSET + k, widths=w, stepout=stepout, **kw) for k in range(NCHAINS)]
    _compare(job, chains, 40, 5, 1, False)


def test_hmc_on_the_rats_hierarchical_model():
    t = cases.rats_target()
    lt, grad = M.hier_normal_target(t.Y, t.xc, t.prior_prec, t.gamma_a, t.gamma_b)
    kw = dict(nsteps=60, burnin=10, thinning=1)
    job = O.OracleJob(sampler=L.SAMPLER_HMC, target_kind=L.TARGET_HIER_NORMAL, nchains=NCHAINS, ndims=t.ndims, leapstep=0.02, nleaps=8, hier_Y=t.Y, hier_xc=t.xc,
                      hier_prior_prec=t.prior_prec, hier_gamma_a=t.gamma_a, hier_gamma_b=t.gamma_b, seed=SEED, chain_offset=OFFSET, **kw)
    x0 = t.least_squares_start()[None, :] + 0.05 * np.random.default_rng(13).standard_normal((NCHAINS, t.ndims))
    assert job.set_state(x0) == 0
    chains = [M.Chain("hmc", lt, grad, x0[k], SEED, OFFSET + k, leapstep=0.02, nleaps=8, **kw) for k in range(NCHAINS)]
    _compare(job, chains, 60, 10, 1, True)
    assert 0.3 < np.mean([np.mean(c.accepts) for c in chains]) < 1.0


def test_mala_on_the_swiss_logistic_regression():
    X, y = cases.swiss_data()
    lt, grad = M.logistic_target(X, y, 100.0)
    kw = dict(nsteps=150, burnin=20, thinning=1)
    job = O.OracleJob(sampler=L.SAMPLER_MALA, target_kind=L.TARGET_LOGISTIC, nchains=NCHAINS, ndims=4, driftstep=0.1, logit_X=X, logit_y=y, logit_lambda=100.0,
                      seed=SEED, chain_offset=OFFSET, **kw)
    x0 = np.array([5.1, -0.9, 8.2, -4.5])[None, :] + 0.1 * np.random.default_rng(8).standard_normal((NCHAINS, 4))
    assert job.set_state(x0) == 0
    chains = [M.Chain("mala", lt, grad, x0[k], SEED, OFFSET + k, driftstep=0.1, **kw) for k in range(NCHAINS)]
    _compare(job, chains, 150, 20, 1, True)


@pytest.mark.parametrize("literal", [False, True])
def test_long_runs_against_the_mirror(literal):
    """the mirror used to be compared on 3 chains x <= 150 transitions.  Here: 16 chains x 2,000 transitions of HMC (L = 10, eps = 0.1 —
    cfg 3's sampler) on a dense 12 x 12 precision with a mean, and 16 chains x 2,000 of MALA on the swiss logistic regression (cfg 4), against the C
    oracle in its shipped arithmetic AND in its literal-Julia mode (unmerged unfused leapfrog, abs2/step, two exponentials — the arithmetic the
    mirror itself uses): every one of the 64,000 accept decisions identical, final states equal to 1e-9 (shipped) / 1e-10 (literal; what is left
    are the mirror's libm transcendentals and NumPy's summation order)."""
    lib = O.load()
    n = 16
    lib.ko_set_literal(1 if literal else 0)
    try:
        d = 12
        rng = np.random.default_rng(31)
        a = rng.standard_normal((d, d)); P = a @ a.T / d + np.eye(d); mu = rng.standard_normal(d)
        lt, grad = M.dense_target(P, mu, -0.5)
        kw = dict(nsteps=2000, burnin=400, thinning=1)
        job = O.OracleJob(sampler=L.SAMPLER_HMC, target_kind=L.TARGET_GAUSS_DENSE, nchains=n, ndims=d, leapstep=0.1, nleaps=10, gauss_prec=P, gauss_mu=mu,
                          gauss_const=-0.5, seed=SEED, chain_offset=OFFSET, **kw)
        x0 = mu[None, :] + rng.standard_normal((n, d))
        assert job.set_state(x0) == 0
        chains = [M.Chain("hmc", lt, grad, x0[k], SEED, OFFSET + k, leapstep=0.1, nleaps=10, **kw) for k in range(n)]
        assert job.run(2000) == 0
        for k, c in enumerate(chains):
            c.run(2000)
            assert np.array_equal(np.asarray(job.accept[:, k], bool), np.array(c.accepts)), ("HMC accept decisions differ", k)
            assert np.allclose(job.X[k], c.x, rtol=1e-10 if literal else 1e-9, atol=1e-11), k
            assert np.allclose(job.sum[k] / 1600, np.mean(c.saved, axis=0), rtol=1e-9, atol=1e-11)
        assert 0.8 < job.accept.mean() <= 1.0
        X, y = cases.swiss_data()
        lt, grad = M.logistic_target(X, y, 100.0)
        kw = dict(nsteps=2000, burnin=100, thinning=1)
        job = O.OracleJob(sampler=L.SAMPLER_MALA, target_kind=L.TARGET_LOGISTIC, nchains=n, ndims=4, driftstep=0.1, logit_X=X, logit_y=y, logit_lambda=100.0,
                          seed=SEED, chain_offset=OFFSET, **kw)
        x0 = np.array([5.1, -0.9, 8.2, -4.5])[None, :] + 0.1 * np.random.default_rng(8).standard_normal((n, 4))
        assert job.set_state(x0) == 0
        chains = [M.Chain("mala", lt, grad, x0[k], SEED, OFFSET + k, driftstep=0.1, **kw) for k in range(n)]
        assert job.run(2000) == 0
        for k, c in enumerate(chains):
            c.run(2000)
            assert np.array_equal(np.asarray(job.accept[:, k], bool), np.array(c.accepts)), ("MALA accept decisions differ", k)
            assert np.allclose(job.X[k], c.x, rtol=1e-9, atol=1e-11), k
    finally:
        lib.ko_set_literal(0)
