"""How far is the shipped arithmetic from the literal Julia arithmetic?  

The oracle the kernels are compared with shares three deliberate deviations with them: (7) the merged fma
leapfrog (/root/reference/src/samplers/samplers.jl:122-134 writes four unmerged, unfused updates per step), (8) the slice sampler's
comparisons on a diagonal Gaussian in difference form (src/samplers/iterate/SliceSampler.jl:66-95 evaluates the whole log-target per probe), (2) MALA's
`0.5*(abs2.(...)/step)` evaluated as `abs2(.)*(0.5/step)` (src/samplers/iterate/MALA.jl:88-92), (6) the logistic rows' two exponentials
taken from one (doc/examples/swiss/MALA/analytical.jl:13,17).  `ko_set_literal(1)` takes all four back.  These tests run the SAME jobs
on the SAME stream in both modes at the BASELINE configurations' trajectory lengths — 64 chains x 2,000 transitions — and measure
  * the first transition at which any state bit differs, and the first at which an accept decision differs,
  * the Hamming fraction of the accept masks (differing decisions / all decisions),
  * |delta mean| / sd and |delta var| / var of the pooled post-burn-in samples, per dimension (maximum),
and assert what can be asserted of the moments (see _check): identical decisions and 1e-9 agreement on every Gaussian / logistic job,
agreement as two independent runs of one chain law on the hierarchical model, where decisions do flip.  `python tests/test_literal_arithmetic.py` prints the report
(committed as profiles/r4_literal_vs_shipped.json, quoted in DESIGN.md section 2).
CPU only (-m "not gpu"); the oracle is the thing under test here, not a stand-in for the product.
"""
import json
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent)); sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import cases  # noqa: E402
import oracle_ffi as O  # noqa: E402
import klara_jl_amd as K  # noqa: E402
from klara_jl_amd import _lib as L  # noqa: E402

NCHAINS, NSTEPS, BURNIN = 64, 2000, 1000


def _jobs():
    """name -> (case dict, what the configuration is)"""
    rng = np.random.default_rng(20260927)
    rats = cases.rats_target()
    X, y = cases.swiss_data()
    return {
        # BASELINE cfg 3: HMC eps = 0.1, L = 10, dense compound-symmetric precision, D = 100
        "cfg3_hmc_dense_L10": dict(sampler=L.SAMPLER_HMC, target=K.GaussDenseTarget(cases.compound_symmetric_precision(100)), leapstep=0.1, nleaps=10,
                                   x0=rng.standard_normal((NCHAINS, 100))),
        # BASELINE cfg 5: HMC L = 32 on the rats model, pooled AcceptanceRate(0.65, period 100) over burn-in 1000
        "cfg5_hmc_rats_L32": dict(sampler=L.SAMPLER_HMC, target=rats, leapstep=0.02, nleaps=32, tuner=L.TUNER_ACCEPT_RATE, tuner_mode=L.TUNE_POOLED,
                                  targetrate=0.65, period=100, x0=rats.least_squares_start()[None, :] + 0.05 * rng.standard_normal((NCHAINS, rats.ndims))),
        # ... the same without the pooled tuner (which couples the chains: one different accept decision changes the pooled rate, hence the
        # step of EVERY chain from the next tuning event on): chains stay independent, divergence is per chain
        "cfg5_hmc_rats_L32_untuned": dict(sampler=L.SAMPLER_HMC, target=rats, leapstep=0.02, nleaps=32,
                                          x0=rats.least_squares_start()[None, :] + 0.05 * rng.standard_normal((NCHAINS, rats.ndims))),
        # north_star's "100-dim Gaussian HMC": README target, eps = 0.1, L = 10
        "hmc_iso_d100_L10": dict(sampler=L.SAMPLER_HMC, target=K.GaussDiagTarget.negdot(100), leapstep=0.1, nleaps=10, x0=rng.standard_normal((NCHAINS, 100))),
        # BASELINE cfg 2's sampler at a drift step that moves (0.9 accepts 0.4 %): abs2(.)/h against abs2(.)*(0.5/h)
        "mala_iso_d100_h0.02": dict(sampler=L.SAMPLER_MALA, target=K.GaussDiagTarget.negdot(100), driftstep=0.02, x0=rng.standard_normal((NCHAINS, 100))),
        # BASELINE cfg 4: MALA h = 0.1 on the swiss logistic regression: two exponentials per row against one, and the quotient
        "cfg4_mala_swiss": dict(sampler=L.SAMPLER_MALA, target=K.LogisticTarget(X, y, 100.0), driftstep=0.1,
                                x0=np.array([5.1, -0.9, 8.2, -4.5]) * 0.0 + 0.1 * rng.standard_normal((NCHAINS, 4))),
        # the slice sampler on a diagonal Gaussian in the pair-transposed layout (deviation (8)): comparisons in difference form,
        # t_i(current) - t_i(candidate) > log(rand()), against a full evaluation of every probe compared with log(rand()) + lt (SliceSampler.jl:66-95)
        "slice_mvnormal_d40": dict(sampler=L.SAMPLER_SLICE, target=K.GaussDiagTarget.mvnormal(np.linspace(-2.0, 3.0, 40), np.linspace(0.5, 2.0, 40)),
                                   slice_widths=np.linspace(0.6, 1.8, 40), slice_stepout=True, x0=rng.standard_normal((NCHAINS, 40))),
    }


def _run(case, literal):
    lib = O.load()
    kw = cases.oracle_kwargs(dict(case, nchains=NCHAINS, nsteps=NSTEPS, burnin=BURNIN, name="literal"))
    job = O.OracleJob(**kw, want_hist=True)
    lib.ko_set_literal(1 if literal else 0)
    try:
        assert lib.ko_get_literal() == (1 if literal else 0)
        assert job.set_state(case["x0"]) == 0
        first_x = None
        # (state trajectories are compared through the saved history after burn-in and through the accept masks over all transitions)
        assert job.run(NSTEPS) == 0
    finally:
        lib.ko_set_literal(0)
    return job


def compare(name, case):
    a, b = _run(case, False), _run(case, True)             # shipped, literal
    diff = a.accept != b.accept                              # (steps x chains)
    per_step = diff.any(axis=1)
    first_acc = int(np.argmax(per_step)) + 1 if per_step.any() else None
    hist_diff = (a.hist != b.hist).any(axis=(1, 2))          # saved steps (post burn-in)
    first_hist = int(np.argmax(hist_diff)) + 1 + BURNIN if hist_diff.any() else None
    va, vb = a.hist.reshape(-1, a.D), b.hist.reshape(-1, b.D)
    sd = va.std(axis=0)
    dmean = np.abs(va.mean(axis=0) - vb.mean(axis=0)) / sd
    dvar = np.abs(va.var(axis=0) - vb.var(axis=0)) / va.var(axis=0)
    maxrel = float(np.max(np.abs(a.hist - b.hist) / (np.abs(a.hist) + sd[None, None, :])))
    # the run's own Monte-Carlo standard error of a pooled mean, in posterior sd's, from the 64 chain means
    cm, cmb = a.hist.mean(axis=0), b.hist.mean(axis=0)       # (chains x D)
    se_d = cm.std(axis=0, ddof=1) / np.sqrt(NCHAINS)
    se = float(np.median(se_d / sd))
    # z-score of the difference of the two runs' pooled means as if they were INDEPENDENT runs (they are, once trajectories part)
    z = np.abs(va.mean(axis=0) - vb.mean(axis=0)) / np.sqrt(se_d ** 2 + (cmb.std(axis=0, ddof=1) / np.sqrt(NCHAINS)) ** 2)
    chains_diverged = int((a.accept != b.accept).any(axis=0).sum())
    return {"job": name, "chains": NCHAINS, "transitions": NSTEPS, "burnin": BURNIN, "acceptance_shipped": float(a.accept.mean()),
            "first_transition_with_a_different_accept_decision": first_acc,
            "first_saved_transition_with_a_different_state_bit": first_hist,
            "accept_mask_hamming_fraction": float(diff.mean()), "chains_with_a_different_accept_decision": chains_diverged,
            "max_z_of_delta_mean_as_independent_runs": float(z.max()),
            "max_abs_delta_mean_over_sd": float(dmean.max()), "max_rel_delta_var": float(dvar.max()),
            "max_relative_state_difference_over_saved_steps": maxrel, "monte_carlo_se_of_pooled_mean_over_sd": se}


def _check(r):
    """north_star asks for moments within 1e-3.  Where the two arithmetics take the same accept decisions throughout (every job but the
    hierarchical model) they must agree far inside that — 1e-9 of a posterior sd — i.e. the deviation is invisible at any run length.  Where a
    decision flips (the rats model: exp() in the target and 32-step trajectories amplify a 1-ulp difference until some |u - a| margin is
    crossed), the two runs are from then on different realisations of the SAME chain law, and the only meaningful statement is statistical: the
    pooled means agree like two independent runs do (|z| < 4.5 over the 65 dimensions), with the difference of the order of the run's own
    standard error — a 1e-3 statement would need ~50 x 64 x 2,000 transitions from the CPU oracle and is made where it can be, by the GPU
    moment tests against the analytic / published posterior (tests/test_gpu_workloads.py, tests/test_gpu_models.py)."""
    if r["accept_mask_hamming_fraction"] == 0.0:
        assert r["max_abs_delta_mean_over_sd"] < 1e-9 and r["max_rel_delta_var"] < 1e-9, r
        assert r["max_relative_state_difference_over_saved_steps"] < 1e-9, r
    else:
        assert r["max_z_of_delta_mean_as_independent_runs"] < 4.5, r
        assert r["max_abs_delta_mean_over_sd"] < 5.0 * r["monte_carlo_se_of_pooled_mean_over_sd"] * 2 ** 0.5 + 1e-3, r
        # (a pair of diverged realisations differs in 2 a (1 - a) ~ 0.49 of its decisions at this acceptance; the fraction over the whole run
        # is that times the share of the run spent after a chain's first flip — it depends on the stream, 0.08-0.12 for the seeds tried)
        assert r["accept_mask_hamming_fraction"] < 0.25, r


def test_literal_leapfrog_against_merged_at_cfg3_length():
    r = compare("cfg3_hmc_dense_L10", _jobs()["cfg3_hmc_dense_L10"])
    _check(r)
    assert r["first_saved_transition_with_a_different_state_bit"] is not None      # the modes really differ (<= 1 ulp per update, from the first step)
    assert r["accept_mask_hamming_fraction"] == 0.0


def test_literal_leapfrog_against_merged_at_cfg5_length():
    jobs = _jobs()
    for name in ("cfg5_hmc_rats_L32", "cfg5_hmc_rats_L32_untuned"):
        r = compare(name, jobs[name])
        _check(r)
        assert r["first_saved_transition_with_a_different_state_bit"] is not None


def test_literal_mala_quotient_and_two_exponential_logistic():
    for name in ("mala_iso_d100_h0.02", "cfg4_mala_swiss", "hmc_iso_d100_L10"):
        _check(compare(name, _jobs()[name]))


def test_literal_mode_is_off_by_default_and_changes_bits():
    lib = O.load()
    assert lib.ko_get_literal() == 0
    case = cases.make_case("hmc_d100")
    outs = []
    for lit in (0, 1, 0):
        job = O.OracleJob(**cases.oracle_kwargs(case))
        lib.ko_set_literal(lit)
        try:
            job.init_state_normal(); job.run(case["nsteps"])
        finally:
            lib.ko_set_literal(0)
        outs.append(job.X.copy())
    assert np.array_equal(outs[0], outs[2]) and not np.array_equal(outs[0], outs[1])
    assert np.allclose(outs[0], outs[1], rtol=1e-9, atol=1e-12)


if __name__ == "__main__":
    print(json.dumps([compare(n, c) for n, c in _jobs().items()], indent=1))


def test_slice_difference_form_against_full_evaluations():
    r = compare("slice_mvnormal_d40", _jobs()["slice_mvnormal_d40"])
    assert r["first_saved_transition_with_a_different_state_bit"] is None, r        # 64 x 2,000 x 40 coordinate updates: the same points, bit for bit
    _check(r)
