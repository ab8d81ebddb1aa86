"""GPU parity tests (-m gpu): libklara_hip.so through its C ABI vs the CPU oracle and the golden vectors.

Bar: bit-exact accept masks, states, log-targets, gradients, per-chain sums and tuner state (the
arithmetic is IEEE-identical by construction: shared Philox stream, shared deterministic log/exp/sincos,
same summation order); pooled moments within the tolerance stated in each test.
"""
import ctypes as C

import os

import numpy as np
import pytest

import cases
import oracle_ffi as O
import klara_jl_amd as K
from klara_jl_amd import _lib as L

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("gpu_required")]


# ------------------------------------------------------------------ primitives
def test_device_math_bit_exact(klib):
    rng = np.random.default_rng(7)
    n = 1 << 18
    sets = {
        0: np.concatenate([rng.random(n), np.exp(rng.uniform(-700, 700, n)), [1.0, 2.0 ** -53, 5e-324]]),
        1: np.concatenate([rng.uniform(-745, 709, n), rng.uniform(-2, 2, n), [0.0, -800.0, 800.0]]),
        2: rng.random(n), 3: rng.random(n),
        4: np.concatenate([rng.random(n) * 1e3, np.exp(rng.uniform(-700, 700, n))]),
        5: np.exp(rng.uniform(-300, 300, n)),
        6: np.concatenate([rng.uniform(-7, 7, n), [0.0, np.inf, -np.inf]]),
        7: np.concatenate([(rng.integers(0, 2 ** 52, n).astype(np.float64) + 0.5) * 2.0 ** -52, np.exp(rng.uniform(-700, 700, n)),
                           [1.0, 2.0 ** -53, 1.0 - 2.0 ** -53, 0.6875, 1.375]]),
        # the Box-Muller radicand -2 log(u): every magnitude it can take, plus values one ulp either side of exact squares
        8: np.concatenate([-2.0 * np.log((rng.integers(0, 2 ** 52, 4 * n).astype(np.float64) + 0.5) * 2.0 ** -52),
                           np.exp(rng.uniform(np.log(2e-16), np.log(74.0), 2 * n)),
                           np.nextafter(np.arange(1, 2000, dtype=np.float64) ** 2 / 64.0, 0.0),
                           np.nextafter(np.arange(1, 2000, dtype=np.float64) ** 2 / 64.0, 100.0),
                           np.arange(1, 2000, dtype=np.float64) ** 2 / 64.0, [2.2e-16, 73.47]]),
        # the logistic rows' functions: exp(-a), the softplus / logistic pair, log on [1, 2]
        9: np.concatenate([rng.uniform(0, 750, n), rng.uniform(0, 40, n), [0.0, 708.0, 708.5, 1e9, np.inf]]),
        10: np.concatenate([rng.uniform(-750, 750, n), rng.uniform(-40, 40, n), [0.0, -0.0, np.nan, 800.0, -800.0]]),
        11: np.concatenate([rng.uniform(-750, 750, n), rng.uniform(-40, 40, n), [0.0, -0.0, np.nan, 800.0, -800.0]]),
        12: np.concatenate([1.0 + rng.random(n), 1.0 + rng.random(n) * 2.0 ** -7, [1.0, 2.0, 1.0078125, 2.0 - 2.0 ** -52]]),
    }
    for op, x in sets.items():
        x = np.ascontiguousarray(x)
        y = np.ascontiguousarray(np.exp(rng.uniform(-300, 300, x.size)))
        out = np.empty_like(x)
        L.check(klib.klara_selftest_math(0, op, x.size, x.ctypes.data, y.ctypes.data, out.ctypes.data), "selftest_math")
        ref = O.math_op(op, x, y)
        assert np.array_equal(out.view(np.uint64), ref.view(np.uint64)), f"op {op}: device math differs from host"


def test_stream_is_rocrand_philox(klib):
    """In-kernel generator == CPU oracle == rocRAND's device philox4x32_10 engine
    (seed, subsequence = global chain id, offset = 4 * (transition << 24 | slot))."""
    seed, chain, t = 20260927, (1 << 33) + 17, 12345
    first = t << 24
    out = np.zeros((64, 4), np.uint32)
    L.check(klib.klara_selftest_rocrand_blocks(0, seed, chain, first, 64, out.ctypes.data), "selftest_rocrand")
    assert np.array_equal(out, O.stream_blocks(seed, chain, t, range(64)))


def test_mfma_f64_order(klib):
    """v_mfma_f64_16x16x4_f64 accumulates each output as one fma chain over k ascending from C —
    the order the oracle's dense gradient uses."""
    rng = np.random.default_rng(3)
    A = rng.standard_normal((16, 4)) * np.exp(rng.uniform(-20, 20, (16, 4)))
    B = rng.standard_normal((4, 16)) * np.exp(rng.uniform(-20, 20, (4, 16)))
    Cm = rng.standard_normal((16, 16))
    D = np.empty((16, 16))
    L.check(klib.klara_selftest_mfma_f64(0, A.ctypes.data, B.ctypes.data, Cm.ctypes.data, D.ctypes.data), "selftest_mfma")
    import math
    ref = np.empty((16, 16))
    for i in range(16):
        for j in range(16):
            acc = Cm[i, j]
            for k in range(4):
                acc = math.fma(A[i, k], B[k, j], acc) if hasattr(math, "fma") else _fma(A[i, k], B[k, j], acc)
            ref[i, j] = acc
    assert np.array_equal(D, ref)


def test_mfma_f64_4x4x4_order(klib):
    """v_mfma_f64_4x4x4_4b (the dense kernel's 4-row tail tile): operand lanes A_b[i][k] = 16k+4b+i, B_b[k][j] = 16k+4b+j,
    result lane D_b[i][j] = 16i+4b+j, and every output element is ONE fma chain over k ascending starting from C —
    the order the oracle's dense gradient uses for every row."""
    import math
    rng = np.random.default_rng(11)
    for trial in range(20):
        A = rng.standard_normal(64) * 10.0 ** rng.integers(-8, 8, 64)
        B = rng.standard_normal(64) * 10.0 ** rng.integers(-8, 8, 64)
        Cm = rng.standard_normal(64) * 10.0 ** rng.integers(-8, 8, 64)
        D = np.empty(64)
        L.check(klib.klara_selftest_mfma_f64_4x4x4(0, A.ctypes.data, B.ctypes.data, Cm.ctypes.data, D.ctypes.data), "selftest_mfma4")
        ref = np.empty(64)
        for b in range(4):
            for i in range(4):
                for j in range(4):
                    acc = Cm[16 * i + 4 * b + j]
                    for k in range(4):
                        x, y = A[16 * k + 4 * b + i], B[16 * k + 4 * b + j]
                        acc = math.fma(x, y, acc) if hasattr(math, "fma") else _fma(x, y, acc)
                    ref[16 * i + 4 * b + j] = acc
        assert np.array_equal(D, ref), trial


def _fma(a, b, c):
    # exact fma via the oracle's libm-free path: use numpy longdouble (64-bit mantissa) is not exact in
    # general, so go through fractions
    from fractions import Fraction
    return float(Fraction(a) * Fraction(b) + Fraction(c))


# ------------------------------------------------------------------ sampler parity
def _run_pair(case, splits=None, spl=0, chain_offset=0):
    eng = K.Engine(**cases.engine_kwargs(case, steps_per_launch=spl, chain_offset=chain_offset))
    layout = eng.layout()
    job = O.OracleJob(**cases.oracle_kwargs(case, layout=layout, chain_offset=chain_offset))
    if case["x0"] is None:
        eng.init_state_normal(); st = job.init_state_normal()
    else:
        eng.set_state(case["x0"]); st = job.set_state(case["x0"])
    assert st == 0
    x, lt, g = eng.state()
    assert np.array_equal(x, job.X), "initial values differ"
    assert np.array_equal(lt, job.LT), "initial log-target differs"
    if case["sampler"] in (L.SAMPLER_MALA, L.SAMPLER_HMC):
        assert np.array_equal(g, job.G), "initial gradient differs"
    for n in (splits or [case["nsteps"]]):
        eng.run(n)
        assert job.run(n) == 0
    return eng, job


def _assert_same(eng, job, case):
    name = case["name"]
    mask = eng.accept_mask()
    assert mask.shape == job.accept.shape
    assert np.array_equal(mask, job.accept), f"{name}: accept mask differs at {np.argwhere(mask != job.accept)[:5]}"
    x, lt, g = eng.state()
    assert np.array_equal(x, job.X), f"{name}: values differ"
    assert np.array_equal(lt, job.LT), f"{name}: log-target differs"
    if case["sampler"] in (L.SAMPLER_MALA, L.SAMPLER_HMC):
        assert np.array_equal(g, job.G), f"{name}: gradient differs"
    s, q, nsaved = eng.chain_sums()
    assert np.array_equal(s, job.sum) and np.array_equal(q, job.sumsq), f"{name}: per-chain sums differ"
    assert nsaved == len(range(case.get("burnin", 0) + 1, case["nsteps"] + 1, case.get("thinning", 1)))
    na, nst = eng.accept_counts()
    assert np.array_equal(na, job.naccept) and nst == case["nsteps"]
    step, acc, prop, tot = eng.tune()
    if case.get("tuner_mode", 0) == L.TUNE_POOLED:
        assert step[0] == job.step[0] and acc[0] == job.accepted[0] and prop[0] == job.proposed[0] and tot[0] == job.totproposed[0]
    else:
        assert np.array_equal(step, job.step, equal_nan=True), f"{name}: tuned step differs"
        assert np.array_equal(acc, job.accepted) and np.array_equal(prop, job.proposed) and np.array_equal(tot, job.totproposed)
    if case.get("tuner", 0) == L.TUNER_DUAL_AVERAGING:
        eb, hb = eng.dual_averaging()
        assert np.array_equal(eb, job.da_epsbar) and np.array_equal(hb, job.da_hbar), f"{name}: dual-averaging state differs"
    # pooled summaries: device tree-sum vs numpy sum — tolerance 1e-12 relative (different association)
    ps, pq, pna, pnt, pns = eng.pooled_summaries()
    assert np.allclose(ps, job.sum.sum(0), rtol=1e-12, atol=1e-9) and np.allclose(pq, job.sumsq.sum(0), rtol=1e-12)
    assert pna == int(job.naccept.sum()) and pnt == case["nsteps"] * case["nchains"] and pns == nsaved


@pytest.mark.parametrize("mode", ["one_launch", "launch_per_transition", "mixed"])
@pytest.mark.parametrize("name", cases.DIAGT_CASES)
def test_parity_pair_transposed_layout(name, mode):
    """Layout kind 3 (klara_diagt.h): 8 lanes per chain, element pairs dealt round-robin.  Accept mask, state,
    log-target, gradient and accept counts are bit-identical to the oracle run with the same summation order, whether
    the transitions go in one launch (multi-step kernel), one launch each (single-transition kernel) or a mix."""
    case = cases.make_case(name)
    n = case["nsteps"]
    eng = K.Engine(**cases.engine_kwargs(case, monitor=L.MON_ACCEPT, steps_per_launch={"one_launch": 0, "launch_per_transition": 1, "mixed": 7}[mode]))
    layout = eng.layout()
    assert layout[0] == 3 and layout[1] == 8, layout       # (MH / MALA up to D = 104 run on 4-lane kernels that sum in this 8-lane order)
    assert tuple(layout) == tuple(O.default_layout(case["target"].kind, case["target"].ndims, sampler=case["sampler"], summaries=False))
    job = O.OracleJob(**cases.oracle_kwargs(case, layout=layout))
    eng.init_state_normal(); assert job.init_state_normal() == 0
    x, lt, g = eng.state()
    assert np.array_equal(x, job.X) and np.array_equal(lt, job.LT)
    if case["sampler"] != L.SAMPLER_MH:
        assert np.array_equal(g, job.G), "initial gradient differs"
    for k in ([n] if mode != "mixed" else [1, n - 4, 3]):
        eng.run(k); assert job.run(k) == 0
    mask = eng.accept_mask()
    assert np.array_equal(mask, job.accept), f"{name}: accept mask differs at {np.argwhere(mask != job.accept)[:5]}"
    assert 0 < mask.sum() < mask.size or name == "dt_mala_d100"        # both outcomes occur (drift 0.9 at D=100 rarely accepts)
    x, lt, g = eng.state()
    assert np.array_equal(x, job.X) and np.array_equal(lt, job.LT), f"{name}: state differs"
    if case["sampler"] != L.SAMPLER_MH:
        assert np.array_equal(g, job.G), f"{name}: gradient differs"
    na, nst = eng.accept_counts()
    assert np.array_equal(na, job.naccept) and nst == n
    eng.close()


def test_group_layout_dimension_sweep():
    """Every dimension 1..72 and a few larger ones on the group layout (E = 2 / 4 / 8 elements per lane, 1..64 lanes per
    chain), MALA with a per-dimension target: state, log-target, gradient and accept counts bit-identical to the oracle."""
    os.environ["KLARA_LAYOUT_KIND"] = "0"
    try:
        for d in list(range(1, 73)) + [100, 127, 128, 129, 200, 255, 256, 257, 400, 511, 512]:
            case = dict(sampler=L.SAMPLER_MALA, target=K.GaussDiagTarget.mvnormal(np.linspace(-1, 1, d), np.linspace(0.7, 1.5, d)), nchains=5,
                        nsteps=6, burnin=0, driftstep=0.3 if d < 130 else 0.1, x0=None, seed=d, name=f"sweep_d{d}")
            eng, job = _run_pair(case, spl=2)
            assert eng.layout()[0] == 0
            x, lt, g = eng.state()
            assert np.array_equal(x, job.X) and np.array_equal(lt, job.LT) and np.array_equal(g, job.G), d
            assert np.array_equal(eng.accept_counts()[0], job.naccept), d
            eng.close()
    finally:
        del os.environ["KLARA_LAYOUT_KIND"]


@pytest.mark.parametrize("sampler,kw,step", [(L.SAMPLER_MALA, dict(driftstep=0.3), 1), (L.SAMPLER_HMC, dict(leapstep=0.2, nleaps=3), 3),
                                             (L.SAMPLER_MH, None, 3), (L.SAMPLER_SLICE, "slice", 7)])
def test_pair_transposed_dimension_sweep(sampler, kw, step):
    """Dimensions 17..128 (the 4-lane kernels for MH / MALA up to 104 at the odd ones — sparse_moves = 1 —, the 8-lane kernels
    otherwise), odd ones included, on the pair-transposed layout (every one for MALA, every
    third for HMC and MH, every seventh for the slice sampler), then 129..512 (16 and 32 lanes per chain) in coarser steps, then 513..1024 (round 6: 64 lanes per
    chain — one chain per wavefront) at six sizes:
    with and without padding pairs / a half pair, i.e. every way of obtaining the accept draw and of storing the last pair."""
    wide = {1: 3, 3: 13, 7: 61}[step]
    for d in list(range(17, 129, step)) + list(range(129, 513, wide)) + [256, 257, 511, 512, 513, 640, 641, 777, 1023, 1024]:
        skw = dict(slice_widths=np.full(d, 1.5)) if kw == "slice" else (kw if kw is not None else dict(mh_sigma=np.full(d, 0.2)))
        case = dict(sampler=sampler, target=K.GaussDiagTarget.mvnormal(np.linspace(-1, 1, d), np.linspace(0.7, 1.5, d)), nchains=11,
                    nsteps=6, burnin=0, x0=None, seed=d, name=f"sweep3_d{d}", sparse_moves=(1 if d % 2 == 1 else 2), **skw)
        eng, job = _run_pair(case, spl=2)
        lanes = 8 if d <= 128 else 16 if d <= 256 else 32 if d <= 512 else 64       # (the summation order; MH / MALA up to 104 at the odd dimensions run the
        assert eng.layout()[:2] == (3, lanes)                    # 4-lane kernels — sparse_moves = 1 —, which reproduce the 8-lane order)
        if sampler in (L.SAMPLER_MH, L.SAMPLER_MALA) and d <= 104:
            assert tuple(eng.launch_modes()[0]) == ((3, 0, 0) if d % 2 == 1 else (0, 3, 0)), d
        x, lt, g = eng.state()
        assert np.array_equal(x, job.X) and np.array_equal(lt, job.LT), d
        assert sampler == L.SAMPLER_MH or np.array_equal(g, job.G), d
        assert np.array_equal(eng.accept_counts()[0], job.naccept), d
        eng.close()


def test_hmc_on_the_4_lane_kernels_sums_in_the_8_lane_order(monkeypatch):
    """Round 5: unmonitored HMC jobs on the diagonal Gaussian (17 <= D <= 104; non-unit diagonals to D = 72) run on the 4-lanes-per-chain kernels — 12.5 of 13
    pair slots real at D = 100 instead of 6.25 of 7 — whose reductions (the two kinetic energies and the log-target) reproduce the 8-lane order like
    MH / MALA's.  Every third dimension, odd ones included, ragged chain counts, fused and one-transition launches: accept mask, state, log-target and
    gradient bit for bit against the oracle told the 8-lane layout; and the same job pinned to the 8-lane kernels gives the same bits."""
    for d in list(range(17, 105, 3)) + [100, 104]:
        unit = d % 2 == 0 or d > 72
        t = K.GaussDiagTarget.negdot(d) if unit else K.GaussDiagTarget.mvnormal(np.linspace(-1, 1, d), np.linspace(0.7, 1.5, d))
        case = dict(sampler=L.SAMPLER_HMC, target=t, nchains=37, nsteps=9, burnin=0, leapstep=0.15, nleaps=4, x0=None, seed=1000 + d, name=f"hmc4_d{d}")
        outs = []
        for pin8 in (False, True):
            if pin8:
                monkeypatch.setenv("KLARA_DIAGT_NO_Q4_HMC", "1")
            else:
                monkeypatch.delenv("KLARA_DIAGT_NO_Q4_HMC", raising=False)
            eng = K.Engine(**cases.engine_kwargs(case, monitor=L.MON_ACCEPT, steps_per_launch=(1 if d % 5 == 0 else 4)))
            assert eng.layout()[:2] == (3, 8)
            job = O.OracleJob(**cases.oracle_kwargs(case, layout=eng.layout()))
            eng.init_state_normal(); job.init_state_normal()
            eng.run(9); job.run(9)
            n4, n8, _ = eng.launch_modes()[0]
            one = d % 5 == 0                              # (one transition per launch: the 8-lane single-transition kernel runs these)
            # (run(9) at 4 transitions per launch = launches of 4, 4 and 1: the last one is a one-transition launch)
            assert ((n4 == 2 and n8 == 1) if not (pin8 or one) else n4 == 0), (d, pin8, n4, n8)
            x, lt, g = eng.state()
            mask = eng.accept_mask()
            assert np.array_equal(mask, job.accept), (d, pin8)
            assert np.array_equal(x, job.X) and np.array_equal(lt, job.LT) and np.array_equal(g, job.G), (d, pin8)
            outs.append((x, mask))
            eng.close()
        assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1]), d
    monkeypatch.delenv("KLARA_DIAGT_NO_Q4_HMC", raising=False)


@pytest.mark.parametrize("nstreams", [0, 3])
def test_few_lanes_layouts_chain_count_sweep(nstreams):
    """1..20 chains (every fill level of the last 8-chain wavefront group) on layout kinds 3 and 4; with three chain
    partitions requested even when there are fewer groups than partitions."""
    rng = np.random.default_rng(3)
    xc = np.array([-14.0, -7.0, 0.0, 7.0, 14.0]); R = 12
    Y = (240.0 + 15.0 * rng.standard_normal(R))[:, None] + (6.0 + 0.5 * rng.standard_normal(R))[:, None] * xc[None, :] + 6.0 * rng.standard_normal((R, 5))
    ht = K.HierNormalTarget(Y, xc)
    for n in range(1, 21):
        case = dict(sampler=L.SAMPLER_MALA, target=K.GaussDiagTarget.negdot(30), nchains=n, nsteps=7, burnin=0, driftstep=0.3, x0=None,
                    seed=n, name=f"chains{n}")
        eng = K.Engine(**cases.engine_kwargs(case, steps_per_launch=2, nstreams=nstreams))
        job = O.OracleJob(**cases.oracle_kwargs(case, layout=eng.layout()))
        assert eng.layout()[0] == 3
        eng.init_state_normal(); job.init_state_normal(); eng.run(7); job.run(7)
        _assert_same(eng, job, case); eng.close()
        if nstreams == 0:
            x0 = ht.least_squares_start()[None, :] + 0.05 * rng.standard_normal((n, ht.ndims))
            case = dict(sampler=L.SAMPLER_HMC, target=ht, nchains=n, nsteps=5, burnin=0, leapstep=0.01, nleaps=3, x0=x0, seed=n, name=f"hchains{n}")
            eng, job = _run_pair(case, spl=2)
            assert eng.layout()[0] == 4
            _assert_same(eng, job, case); eng.close()


@pytest.mark.parametrize("d", [18, 34, 36, 50, 66, 82, 98, 114, 126])
@pytest.mark.parametrize("sampler", [L.SAMPLER_MALA, L.SAMPLER_HMC])
def test_pair_transposed_every_pairs_per_lane(d, sampler):
    """NP = ceil(D/16) = 2..8, each at a dimension where part of the last pair row is padding (the kernels assume that only
    the LAST pair of a lane can be padding, which holds because NP is exactly the ceiling)."""
    kw = dict(driftstep=0.25) if sampler == L.SAMPLER_MALA else dict(leapstep=0.2, nleaps=3)
    case = dict(sampler=sampler, target=K.GaussDiagTarget.mvnormal(np.linspace(-1, 1, d), np.linspace(0.7, 1.5, d)), nchains=13, nsteps=12,
                burnin=2, x0=None, seed=5, name=f"np_d{d}", sparse_moves=1, **kw)
    eng, job = _run_pair(case, splits=[5, 7], spl=3)
    lanes = eng.layout()[1]                                                        # (untuned MALA up to D = 104 runs the 4-lane kernels here)
    assert eng.layout()[0] == 3 and lanes == 8
    assert (eng.launch_modes()[0][0] > 0) == (sampler == L.SAMPLER_MALA and d <= 104)
    assert eng.layout()[2] == 2 * (((d + 1) // 2 + lanes - 1) // lanes)           # NP = ceil(ceil(d/2) / lanes)
    _assert_same(eng, job, case)
    eng.close()


@pytest.mark.parametrize("nchains", [1, 3, 9])
@pytest.mark.parametrize("sampler,kw", [(L.SAMPLER_MALA, dict(driftstep=0.3)), (L.SAMPLER_HMC, dict(leapstep=0.2, nleaps=4)),
                                        (L.SAMPLER_MH, dict(mh_sigma=np.full(22, 0.3)))])
def test_pair_transposed_tiny_jobs(nchains, sampler, kw):
    """Fewer chains than one wavefront group carries (8), D = 22 (two pairs per lane, the last five of them padding)."""
    case = dict(sampler=sampler, target=K.GaussDiagTarget.mvnormal(np.linspace(-1, 1, 22), np.linspace(0.5, 2, 22)), nchains=nchains,
                nsteps=25, burnin=5, thinning=2, x0=None, seed=7, name="tiny", **kw)
    eng, job = _run_pair(case, splits=[1, 9, 15], spl=4)
    assert eng.layout()[0] == 3
    _assert_same(eng, job, case)
    eng.close()


@pytest.mark.parametrize("R,T,nchains,tuner_kw", [(9, 5, 11, {}), (13, 3, 5, dict(tuner=L.TUNER_ACCEPT_RATE, targetrate=0.7, period=8)),
                                                  (30, 8, 64, dict(tuner=L.TUNER_ACCEPT_RATE, tuner_mode=L.TUNE_POOLED, targetrate=0.65, period=10)),
                                                  (32, 12, 19, {}), (17, 2, 8, {})])
def test_hier_few_lanes_layout_other_unit_counts(R, T, nchains, tuner_kw):
    """Layout kind 4 (klara_hiert.h) on synthetic growth-curve data: unit counts with a partly filled last lane (9, 13, 17, 30)
    and a full one (32), 2 to 12 observations per unit on a covariate that is not centred, ragged chain counts, the
    per-chain and the pooled AcceptanceRate tuner, history of every field."""
    rng = np.random.default_rng(R)
    xc = np.linspace(-14.0, 14.0, T) + (0.0 if T == 5 else 1.5)
    a = 240.0 + 15.0 * rng.standard_normal(R); b = 6.0 + 0.5 * rng.standard_normal(R)
    Y = a[:, None] + b[:, None] * xc[None, :] + 6.0 * rng.standard_normal((R, T))
    t = K.HierNormalTarget(Y, xc)
    x0 = t.least_squares_start()[None, :] + 0.05 * rng.standard_normal((nchains, t.ndims))
    case = dict(sampler=L.SAMPLER_HMC, target=t, nchains=nchains, nsteps=24, burnin=16, thinning=2, leapstep=0.01, nleaps=6, x0=x0,
                seed=99, name=f"hier_R{R}", **tuner_kw)
    eng = K.Engine(**cases.engine_kwargs(case, monitor=L.MON_ACCEPT | L.MON_SUMMARIES | L.MON_HISTORY | L.MON_HIST_LT | L.MON_HIST_GRAD,
                                         steps_per_launch=5))
    assert tuple(eng.layout()) == (4, 8, 8)
    job = O.OracleJob(**cases.oracle_kwargs(case, layout=eng.layout()), want_hist=True)
    eng.set_state(x0); assert job.set_state(x0) == 0
    assert np.array_equal(eng.state()[1], job.LT) and np.array_equal(eng.state()[2], job.G)
    for k in (1, 14, 9):
        eng.run(k); assert job.run(k) == 0
    _assert_same(eng, job, case)
    c = nchains // 2
    ltc, gc = eng.chain_fields(c, logtarget=True, gradlogtarget=True)
    assert np.array_equal(eng.chain(c), job.hist[:, c, :].T)
    assert np.array_equal(ltc, job.hist_lt[:, c]) and np.array_equal(gc, job.hist_g[:, c, :].T)
    eng.close()


def test_c_abi_summary_allreduce_over_rccl_single_rank(klib):
    """klara_comm_* / klara_gather_summaries: the C-ABI form of the one multi-GPU exchange (an RCCL all-reduce of the
    pooled summaries).  One box has one GPU, so the communicator has a single rank: the call sequence, the lazy librccl
    load and the packing are exercised, and the result must equal klara_get_pooled_summaries."""
    case = cases.make_case("mala_d100_small_step")
    eng = K.Engine(**cases.engine_kwargs(case)); eng.init_state_normal(); eng.run(case["nsteps"])
    uid = (C.c_uint8 * 128)()
    L.check(klib.klara_comm_unique_id(uid), "comm_unique_id")
    comm = C.c_void_p()
    L.check(klib.klara_comm_init(C.byref(comm), 1, 0, uid, 0), "comm_init")
    d = case["target"].ndims
    s = np.empty(d); q = np.empty(d)
    na, nt, ns, nc = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
    for _ in range(2):          # collective is repeatable
        L.check(klib.klara_gather_summaries(eng._h, comm, s.ctypes.data, q.ctypes.data, C.byref(na), C.byref(nt), C.byref(ns), C.byref(nc)),
                "gather_summaries")
    ps, pq, pna, pnt, pns = eng.pooled_summaries()
    assert np.array_equal(s, ps) and np.array_equal(q, pq)
    assert (na.value, nt.value, ns.value, nc.value) == (pna, pnt, pns * case["nchains"], case["nchains"])
    L.check(klib.klara_comm_destroy(comm), "comm_destroy")
    eng.close()


@pytest.mark.parametrize("name", ["dt_mala_d100", "dt_mala_d18", "dt_mala_mvnormal_d30", "dt_mh_d100", "dt_mh_mvnormal_d20"])
@pytest.mark.parametrize("spl", [0, 1])
def test_pair_transposed_8_lane_form_of_untuned_mh_mala(name, spl, monkeypatch):
    """Untuned MH / MALA jobs up to D = 104 can run on the 4-lane kernels (always when they keep no running sums); under
    KLARA_DIAGT_NO_Q4 they only ever run the 8-lane kernels (the ones their tuned siblings run), with resident running sums instead
    of atomic folds: same bits — both families sum in the 8-lane order."""
    monkeypatch.setenv("KLARA_DIAGT_NO_Q4", "1")
    case = cases.make_case(name)
    eng, job = _run_pair(case, spl=spl)
    assert eng.layout()[:2] == (3, 8)
    _assert_same(eng, job, case)
    eng.close()


def test_layout_choice_matches_its_mirror():
    """tests/oracle_ffi.default_layout (what the CPU-side golden generator assumes) is the product's choice for every
    dimension and tuner it can meet."""
    for d in list(range(1, 140)) + [200, 256, 257, 400, 512]:
        for sampler, extra in ((L.SAMPLER_MALA, dict(driftstep=0.1)), (L.SAMPLER_MH, dict(mh_sigma=np.ones(d))),
                               (L.SAMPLER_SLICE, dict(slice_widths=np.ones(d)))):
            for tuner in (L.TUNER_VANILLA, L.TUNER_ACCEPT_RATE):
                for mon, sparse in ((0, 0), (L.MON_SUMMARIES, 0), (L.MON_SUMMARIES, 1), (L.MON_SUMMARIES, 2)):
                    e = K.Engine(sampler=sampler, target=K.GaussDiagTarget.negdot(d), nchains=5, nsteps=2, tuner=tuner, targetrate=0.5, monitor=mon,
                                 sparse_moves=sparse, **extra)
                    mirror = O.default_layout(L.TARGET_GAUSS_DIAG, d, sampler=sampler, tuner=tuner, summaries=bool(mon), sparse_moves=sparse)
                    assert tuple(e.layout()) == tuple(mirror), (d, sampler, tuner, mon, sparse)
                    e.close()


def test_pair_transposed_layout_is_optional():
    """D < 17 or D > 512 keep the group layout; KLARA_LAYOUT_KIND=0 forces it."""
    case = cases.make_case("dt_mala_d100")
    e = K.Engine(**cases.engine_kwargs(case)); assert e.layout()[0] == 3; e.close()                  # any monitor
    e = K.Engine(**cases.engine_kwargs(case, monitor=0)); assert e.layout()[0] == 3; e.close()
    e = K.Engine(**cases.engine_kwargs(case, monitor=0, tuner=L.TUNER_ACCEPT_RATE, targetrate=0.5)); assert e.layout()[0] == 3; e.close()
    e = K.Engine(**cases.engine_kwargs(case, monitor=0, verbose=True)); assert e.layout()[0] == 3; e.close()
    e = K.Engine(**cases.engine_kwargs(dict(case, sampler=L.SAMPLER_HMC), monitor=0, tuner=L.TUNER_DUAL_AVERAGING, targetrate=0.65, da_nadapt=10))
    assert e.layout()[0] == 3; e.close()
    e = K.Engine(**cases.engine_kwargs(cases.make_case("mala_d129"), monitor=0)); assert e.layout() == (3, 16, 10); e.close()
    e = K.Engine(**cases.engine_kwargs(cases.make_case("mh_mvnormal_d7"), monitor=0)); assert e.layout()[0] == 0; e.close()
    e = K.Engine(**cases.engine_kwargs(cases.make_case("mala_d300"), monitor=0)); assert e.layout() == (3, 32, 10); e.close()
    e = K.Engine(**cases.engine_kwargs(cases.make_case("mh_d512"), monitor=0)); assert e.layout() == (3, 32, 16); e.close()
    e = K.Engine(**cases.engine_kwargs(cases.make_case("slice_d100_nostepout"), monitor=0)); assert e.layout()[0] == 3; e.close()
    e = K.Engine(**cases.engine_kwargs(cases.make_case("slice_d5"), monitor=0)); assert e.layout()[0] == 0; e.close()
    os.environ["KLARA_LAYOUT_KIND"] = "0"
    try:
        e = K.Engine(**cases.engine_kwargs(case, monitor=0)); assert e.layout()[0] == 0; e.close()
    finally:
        del os.environ["KLARA_LAYOUT_KIND"]


@pytest.mark.parametrize("name", ["hmc_dense_d130_wide", "mh_dense_d130_wide", "hmc_dense_d130_dualavg_wide"])
def test_dense_target_beyond_128_closure_form_still_matches(name, monkeypatch):
    """Round 4 moved HMC / MALA / MH on dense targets of 129..256 dimensions onto the matrix cores (streamed P); the run-time compiled closure form
    they used to take (one chain per lane) remains what D > 256 runs (round 5 moved the slice sampler over too), and KLARA_DENSE_NO_STREAM=1 selects it for every sampler:
    both forms against the oracle in their own summation orders (layout kind 1 on 4 lanes; kind 0 on one lane), bit for bit."""
    case = cases.make_case(name)
    eng, job = _run_pair(case)
    assert eng.layout()[0] == 1 and eng.layout()[2] in (40, 48, 56, 64)
    _assert_same(eng, job, case)
    eng.close()
    monkeypatch.setenv("KLARA_DENSE_NO_STREAM", "1")
    eng, job = _run_pair(case)
    assert eng.layout() == (0, 1, 256)
    _assert_same(eng, job, case)
    eng.close()


@pytest.mark.parametrize("name", sorted(cases.SPLIT_CASES))
def test_dense_target_beyond_256_on_workgroup_split_layout(name):
    """Round 6: dense targets of 257 .. 1024 dimensions (refused in rounds 1-5) run on the matrix cores with the tile of 16 chains on a workgroup of
    4 .. 16 wavefronts that deal the row tiles of P evenly (klara_dense_split.h): layout kind 6, bit for bit against the oracle's kind-6 summation order, in one piece and split into launches."""
    case = cases.make_case(name)
    d = case["target"].ndims
    eng, job = _run_pair(case)
    assert eng.layout() == O.split_dense_layout(d)
    _assert_same(eng, job, case)
    eng.close()
    n = case["nsteps"]
    eng, job = _run_pair(case, splits=[1, n // 3, n - 1 - n // 3], spl=7)
    _assert_same(eng, job, case)
    eng.close()


@pytest.mark.parametrize("name", ["hmc_dense_d98", "hmc_dense_d128", "hmc_dense_d100_mean", "mala_dense_d37_mean", "mh_dense_d20_mean", "hmc_dense_d70_mean_dualavg",
                                  "hmc_dense_d256_stream_tuned", "hmc_dense_d160_stream_pooled", "mala_dense_d200_stream_tuned", "mh_dense_d256_stream_mean",
                                  "mala_dense_d130_stream_pooled", "hmc_dense_d130_dualavg_wide", "slice_dense_d20", "slice_dense_d37_mean", "slice_dense_d130_stream_mean"])
def test_smaller_dense_targets_on_the_split_layout(name, monkeypatch):
    """KLARA_DENSE_SPLIT=1 puts every dense target on the workgroup-split layout (4 wavefronts per tile below D = 257, 0 .. 4 row tiles each): the cases of the LDS-resident and
    streamed layouts again, against the oracle in the kind-6 order."""
    monkeypatch.setenv("KLARA_DENSE_SPLIT", "1")
    case = cases.make_case(name)
    d = case["target"].ndims
    eng, job = _run_pair(case)
    assert eng.layout() == O.split_dense_layout(d)
    _assert_same(eng, job, case)
    eng.close()


@pytest.mark.parametrize("name", ["pair_quartic_slice_d100", "pair_banana_slice_d37", "pair_indexed_slice_d300", "pair_indexed_slice_d40_whole"])
def test_slice_sampler_on_pair_closures_runs_on_the_few_lanes_kernels(name, monkeypatch):
    """Round 6: a pair closure under the slice sampler (rounds 1-4: refused; round 5: summed by the library and run as a whole-vector closure, every probe a full
    evaluation) runs on the pair-transposed layout: a lane updates the coordinates of its own pairs and a probe compares the PAIR's term (oracle: ko_slice_pair_delta).
    Both forms against the oracle in their own orders, bit for bit; the new state's log-target is one full evaluation in the layout's order in both."""
    case = cases.make_case(name)
    d = case["target"].ndims
    eng, job = _run_pair(case)
    assert eng.layout()[0] == 3 and eng.layout()[1] == (8 if d <= 128 else 16 if d <= 256 else 32), eng.layout()
    _assert_same(eng, job, case)
    x_pair = eng.state()[0]
    eng.close()
    n = case["nsteps"]
    eng, job = _run_pair(case, splits=[1, n - 1], spl=3)
    _assert_same(eng, job, case)
    eng.close()
    monkeypatch.setenv("KLARA_PAIR_SLICE_AS_WHOLE", "1")
    eng, job = _run_pair(case)
    assert eng.layout()[0] == 0
    _assert_same(eng, job, case)
    # the two forms draw the same uniforms and differ in how a comparison is rounded (a term against a total): the same states up to the rare flip
    assert np.mean(np.abs(eng.state()[0] - x_pair) < 1e-9) > 0.95
    eng.close()


@pytest.mark.parametrize("extra", [dict(verbose=True, period=4), dict(tuner=L.TUNER_ACCEPT_RATE, targetrate=0.5, period=3)], ids=["verbose", "rate"])
def test_slice_sampler_on_pair_closures_with_counting_tuners_and_histories(extra):
    """The same kernel family with something that counts proposals (the TUNE instantiation) and every saved-sample monitor on: state, sums, tuner state, one chain's
    value and log-target histories against the oracle."""
    c = dict(cases.make_case("pair_quartic_slice_d100"), **extra)
    mon = L.MON_ACCEPT | L.MON_SUMMARIES | L.MON_HISTORY | L.MON_HIST_LT
    eng = K.Engine(**cases.engine_kwargs(c, monitor=mon, steps_per_launch=3))
    assert eng.layout()[0] == 3
    job = O.OracleJob(**cases.oracle_kwargs(c, layout=eng.layout()), want_hist=True)
    eng.set_state(c["x0"]); assert job.set_state(c["x0"]) == 0
    eng.run(c["nsteps"]); assert job.run(c["nsteps"]) == 0
    _assert_same(eng, job, c)
    assert np.array_equal(eng.chain(5), job.hist[:, 5, :].T)
    lt, _ = eng.chain_fields(5, logtarget=True, gradlogtarget=False)
    assert np.array_equal(lt, job.hist_lt[:, 5])
    eng.close()


@pytest.mark.parametrize("name", ["hmc_logit_d20_wide", "mala_logitm_d20", "slice_logitm_d20", "mala_logit_d12_manyrows"])
def test_logistic_beyond_16_parameters_closure_form_still_matches(name, monkeypatch):
    """Round 6 moved the logistic regression beyond 16 parameters (and 9 .. 16 with rows that do not fit the LDS) onto the matrix cores; the run-time compiled
    closure form of rounds 1-5 (one chain per lane, all rows on it) remains behind KLARA_LOGIT_NO_MFMA=1: both forms against the oracle in their own summation
    orders (layout kind 5: row sums over 4 lane-quarters; kind 0 on one lane), bit for bit — and with the SAME gradients (the fma chains are the same)."""
    case = cases.make_case(name)
    eng, job = _run_pair(case)
    assert eng.layout()[0] == 5
    _assert_same(eng, job, case)
    g5 = eng.state()[2]
    eng.close()
    monkeypatch.setenv("KLARA_LOGIT_NO_MFMA", "1")
    eng, job = _run_pair(case)
    assert eng.layout()[:2] == (0, 1)
    _assert_same(eng, job, case)
    eng.close()


# Vanilla / AcceptanceRate jobs on even-D diagonal Gaussians run on the pair-transposed layout by default (HMC on the
# hierarchical target: layout kind 4); the same cases forced
# onto the group layout keep that path covered (both compared with the oracle told the respective summation order)
GROUP_FORCED = [n for n in cases.ALL_CASES if n in ("mh_d100", "mala_d100", "mala_d100_small_step", "hmc_d100", "hmc_d128_full",
                                                   "mala_d20_tuned", "hmc_d100_tuned", "mala_d100_verbose",
                                                   "hmc_rats", "hmc_rats_pooled", "hmc_rats_dualavg", "hmc_d40_dualavg", "hmc_d100_dualavg",
                                                   "slice_d100_nostepout", "slice_d20_stepout", "mala_rats", "mh_rats", "mala_rats_tuned",
                                                   "mala_d129", "mala_d300", "mh_d512")]


@pytest.mark.parametrize("name", GROUP_FORCED)
def test_parity_with_oracle_group_layout_forced(name):
    os.environ["KLARA_LAYOUT_KIND"] = "0"
    try:
        case = cases.make_case(name)
        eng, job = _run_pair(case)
        assert eng.layout()[0] in (0, 2) and eng.layout()[0] == 0
        _assert_same(eng, job, case)
    finally:
        del os.environ["KLARA_LAYOUT_KIND"]


@pytest.mark.parametrize("name", cases.ALL_CASES)
def test_parity_with_oracle(name):
    case = cases.make_case(name)
    eng, job = _run_pair(case)
    _assert_same(eng, job, case)
    eng.close()


@pytest.mark.parametrize("name", cases.GOLDEN_CASES)
def test_parity_with_golden_vectors(name):
    """Same comparison against the committed vectors (independent of the oracle build on this box)."""
    case = cases.make_case(name)
    gold = np.load(cases.GOLDEN / f"{name}.npz")
    dt = name in cases.DIAGT_CASES          # layout kind 3 serves jobs that monitor only the accept mask
    eng = K.Engine(**(cases.engine_kwargs(case, monitor=L.MON_ACCEPT) if dt else cases.engine_kwargs(case)))
    assert list(eng.layout()) == list(gold["layout"])
    eng.set_state(gold["x0"])
    eng.run(case["nsteps"])
    x, lt, g = eng.state()
    assert np.array_equal(eng.accept_mask(), gold["accept"])
    assert np.array_equal(x, gold["x"]) and np.array_equal(lt, gold["lt"])
    assert np.array_equal(eng.accept_counts()[0], gold["naccept"])
    if dt:
        eng.close()
        return
    s, q, _ = eng.chain_sums()
    assert np.array_equal(s, gold["sum"]) and np.array_equal(q, gold["sumsq"])
    step = eng.tune()[0]
    if case.get("tuner_mode", 0) == L.TUNE_POOLED:
        assert step[0] == gold["step"][0]
    else:
        assert np.array_equal(step, gold["step"], equal_nan=True)
    eng.close()


@pytest.mark.parametrize("name,splits,spl", [
    ("mala_d100", [1, 7, 42], 1), ("mala_d100", [50], 5), ("hmc_d100", [13, 17], 4),
    ("hmc_dense_d100", [5, 7], 3), ("mala_d3_tuned", [100, 60, 100], 7), ("hmc_d10_tuned_pooled", [33, 87], 16),
    ("slice_d5", [11, 19], 2), ("mala_swiss", [40], 1), ("hmc_dense_d192_stream_mean", [3, 11], 2), ("hmc_dense_d256_stream_tuned", [17, 23], 6),
    ("mala_logitm_d20", [7, 18], 3), ("hmc_logitm_d40_dualavg", [11, 14], 4), ("mala_logitm_d128_n1100_tuned", [5, 17], 1),
])
def test_launch_splitting_does_not_change_results(name, splits, spl):
    """K transitions per launch / multiple klara_run calls are invisible in the results."""
    case = cases.make_case(name)
    eng, job = _run_pair(case, splits=splits, spl=spl)
    _assert_same(eng, job, case)
    eng.close()


@pytest.mark.parametrize("name,monitor", [("mala_d100", L.MON_ACCEPT | L.MON_SUMMARIES), ("dt_mala_d100_small_step", L.MON_ACCEPT),
                                          ("dt_hmc_d100", L.MON_ACCEPT), ("mala_logitm_d20", L.MON_ACCEPT | L.MON_SUMMARIES),
                                          ("hmc_dense_d300_split", L.MON_ACCEPT | L.MON_SUMMARIES)])
def test_sharding_invariance(name, monitor):
    """Rank r's shard (chain_offset) reproduces the same chains as the single-GPU job — group layout and
    pair-transposed layout (the shard boundary does not fall on a wavefront-group boundary there)."""
    case = cases.make_case(name)
    full = K.Engine(**cases.engine_kwargs(case, monitor=monitor)); full.init_state_normal(); full.run(case["nsteps"])
    xf = full.state()[0]; mf = full.accept_mask()
    for rank in (0, 1, 2):
        off, cnt = K.shard_chains(case["nchains"], rank, 3)
        part = K.Engine(**cases.engine_kwargs(dict(case, nchains=cnt), monitor=monitor, chain_offset=off))
        assert part.layout() == full.layout()
        part.init_state_normal(); part.run(case["nsteps"])
        assert np.array_equal(part.state()[0], xf[off:off + cnt]) and np.array_equal(part.accept_mask(), mf[:, off:off + cnt])
        part.close()
    full.close()


def test_reset_and_error_paths_on_pair_transposed_layout():
    case = cases.make_case("dt_mala_d100_small_step")
    eng = K.Engine(**cases.engine_kwargs(case, monitor=L.MON_ACCEPT)); eng.init_state_normal()
    assert eng.layout()[0] == 3
    x0 = eng.state()[0]
    eng.run(case["nsteps"]); x1 = eng.state()[0]; m1 = eng.accept_mask()
    with pytest.raises(K.KlaraError):
        eng.run(1)                                  # beyond nsteps: the accept-mask buffer is full (KLARA_ERR_STATE)
    eng.set_state(x0); eng.run(case["nsteps"])           # set_state keeps the stream: the job is replayed
    assert np.array_equal(eng.state()[0], x1) and np.array_equal(eng.accept_mask(), m1)
    bad = x0.copy(); bad[5, 7] = np.inf
    with pytest.raises(K.KlaraError) as ei:
        eng.set_state(bad)
    assert ei.value.status == L.ERR_NONFINITE_INIT
    eng.close()


def test_reset_rewinds_the_job():
    """reset(job) / reset(job, x) — BasicMCJob.jl:187-201: sampler / tuner state, counters and monitors rewind; the random stream
    does NOT (the reference's generator keeps advancing), so run -> reset(x0) -> run is an independent replicate from the same
    start: a different trajectory, bit-identical to the oracle on the job's next Philox key (klara_hip.h klara_reset), while
    set_state(x0) replays the first run."""
    case = cases.make_case("hmc_d100_tuned")
    eng = K.Engine(**cases.engine_kwargs(case)); eng.init_state_normal()
    job = O.OracleJob(**cases.oracle_kwargs(case, layout=eng.layout())); job.init_state_normal()
    assert eng.stream_key() == (case["seed"], 0)
    x0 = eng.state()[0]
    eng.run(case["nsteps"]); job.run(case["nsteps"]); x1 = eng.state()[0]; m1 = eng.accept_mask()
    _assert_same(eng, job, case)
    for k, x in ((1, x0), (2, None)):                     # reset(job, x0), then reset(job) from wherever it is
        eng.reset(x); assert job.reset(x) == 0
        assert eng.stream_key() == ((case["seed"] + k * 0x9E3779B97F4A7C15) % 2 ** 64, k)
        assert eng.accept_counts()[1] == 0 and eng.tune()[3][0] == case["period"]          # counters and tuner state rewound
        eng.run(case["nsteps"]); assert job.run(case["nsteps"]) == 0
        _assert_same(eng, job, case)
        assert not np.array_equal(eng.accept_mask(), m1)
    # a reset that fails (non-finite values) leaves the job on its key: no epoch for a reset that did not happen
    bad = x0.copy(); bad[1, 2] = np.inf
    with pytest.raises(K.KlaraError) as ei:
        eng.reset(bad)
    assert ei.value.status == L.ERR_NONFINITE_INIT
    assert eng.stream_key() == ((case["seed"] + 2 * 0x9E3779B97F4A7C15) % 2 ** 64, 2)
    eng.reset(x0); assert job.reset(x0) == 0                                               # ... and the next good one is reset number 3
    assert eng.stream_key() == ((case["seed"] + 3 * 0x9E3779B97F4A7C15) % 2 ** 64, 3)
    eng.run(case["nsteps"]); assert job.run(case["nsteps"]) == 0
    _assert_same(eng, job, case)
    eng.close()
    eng = K.Engine(**cases.engine_kwargs(case)); eng.set_state(x0); eng.run(case["nsteps"])
    assert np.array_equal(eng.state()[0], x1) and np.array_equal(eng.accept_mask(), m1)
    eng.close()


def test_nonfinite_initial_values_are_rejected_by_the_logistic_target():
    """A NaN parameter makes every row's Xp NaN; the rows' softplus / logistic pair stays finite (kd_softplus_logistic_rows) but the
    term Xp * y — and with it the log-target — is NaN, which is what initialize! tests (MALA.jl:83-84, MH.jl:72-85)."""
    X, y = cases.swiss_data()
    for sampler, kw in ((L.SAMPLER_MALA, dict(driftstep=0.1)), (L.SAMPLER_MH, dict(mh_sigma=np.full(4, 0.1)))):
        eng = K.Engine(sampler=sampler, target=K.LogisticTarget(X, y, 100.0), nchains=20, nsteps=10, **kw)
        x = np.zeros((20, 4)); x[7, 2] = np.nan
        with pytest.raises(K.KlaraError) as ei:
            eng.set_state(x)
        assert ei.value.status == L.ERR_NONFINITE_INIT
        x[7, 2] = np.inf
        with pytest.raises(K.KlaraError) as ei:
            eng.set_state(x)
        assert ei.value.status == L.ERR_NONFINITE_INIT
        eng.set_state(np.zeros((20, 4))); eng.run(3)
        eng.close()


def test_nonfinite_initial_values_are_rejected():
    eng = K.Engine(sampler=L.SAMPLER_MALA, target=K.GaussDiagTarget.negdot(3), nchains=5, nsteps=10, driftstep=0.1)
    x = np.zeros((5, 3)); x[3, 1] = np.nan
    with pytest.raises(K.KlaraError) as ei:
        eng.set_state(x)
    assert ei.value.status == L.ERR_NONFINITE_INIT
    with pytest.raises(K.KlaraError):
        eng.run(1)                                     # no state -> KLARA_ERR_STATE
    eng.close()


def test_error_paths():
    """Status codes instead of aborts: call order, capacity, unsupported combinations, slice sampler that cannot move."""
    eng = K.Engine(sampler=L.SAMPLER_MH, target=K.GaussDiagTarget.negdot(2), nchains=4, nsteps=5, mh_sigma=[1.0, 1.0],
                   monitor=L.MON_ACCEPT)
    with pytest.raises(K.KlaraError) as ei:
        eng.run(1)                                             # run before set_state
    assert ei.value.status == L.ERR_STATE
    eng.set_state(np.zeros((4, 2)))
    eng.run(5)
    with pytest.raises(K.KlaraError) as ei:
        eng.run(1)                                             # accept diagnostics are sized for range.nsteps
    assert ei.value.status == L.ERR_STATE
    with pytest.raises(K.KlaraError) as ei:
        eng.chain(0)                                           # history not monitored
    assert ei.value.status == L.ERR_STATE
    eng.close()
    for kw in (dict(sampler=L.SAMPLER_HMC, target=K.GaussDenseTarget(np.eye(1025))),                # the matrix-core layouts end at D = 1,024 (round 6: the workgroup-split layout; 256 before)
               dict(sampler=L.SAMPLER_MALA, target=K.GaussDiagTarget.negdot(4), tuner=L.TUNER_DUAL_AVERAGING,
                    targetrate=0.6, da_nadapt=10),
               dict(sampler=L.SAMPLER_SLICE, slice_widths=np.ones(4), target=K.GaussDiagTarget.negdot(4),
                    tuner=L.TUNER_ACCEPT_RATE, targetrate=0.5, tuner_mode=L.TUNE_POOLED)):
        with pytest.raises(K.KlaraError) as ei:
            K.Engine(nchains=4, nsteps=5, **kw)
        assert ei.value.status == L.ERR_UNSUPPORTED
    rng = np.random.default_rng(3)
    for ndata, ok in ((1434, True), (3686, True), (3687, False)):   # logistic data rows live in LDS: ndata * (E + 1) <= 18432 doubles (beyond 7168 the kernels' LDS limit is raised)
        tgt = K.LogisticTarget(rng.standard_normal((ndata, 4)), (rng.random(ndata) < 0.5).astype(float))
        if ok:
            with K.Engine(sampler=L.SAMPLER_MALA, target=tgt, nchains=8, nsteps=2, driftstep=0.01) as e:
                e.set_state(np.zeros((8, 4))); e.run(2)
                assert np.isfinite(e.state()[1]).all()
        else:
            with pytest.raises(K.KlaraError) as ei:
                K.Engine(sampler=L.SAMPLER_MALA, target=tgt, nchains=8, nsteps=2, driftstep=0.01)
            assert ei.value.status == L.ERR_UNSUPPORTED
    with pytest.raises(K.KlaraError) as ei:                    # gradient history needs a gradient-carrying sampler
        K.Engine(sampler=L.SAMPLER_MH, target=K.GaussDiagTarget.negdot(2), nchains=4, nsteps=5, mh_sigma=[1.0, 1.0],
                 monitor=L.MON_HIST_GRAD)
    assert ei.value.status == L.ERR_INVALID_ARG
    # a flat log-target region at -inf: the slice never contains an acceptable point other than the current one
    w = np.array([1e300, 1.0])
    eng = K.Engine(sampler=L.SAMPLER_SLICE, target=K.GaussDiagTarget(2, w=np.array([1e300, 1.0])), nchains=3, nsteps=2,
                   slice_widths=[1e-300, 1.0], slice_stepout=False)
    eng.set_state(np.zeros((3, 2)))
    try:
        eng.run(2)                                             # either moves or reports KLARA_ERR_SLICE_STUCK — never hangs
    except K.KlaraError as e:
        assert e.status == L.ERR_SLICE_STUCK
    eng.close()
    del w


def test_history_layout_matches_nstate():
    """:destination=>:nstate — value is (D x npoststeps) per chain, column i = i-th saved step
    (test/ParameterNStates.jl:139-146,176-185; save rule BasicMCJob.jl:226-231)."""
    case = cases.make_case("mala_d100_small_step")
    eng = K.Engine(**cases.engine_kwargs(case, monitor=L.MON_HISTORY | L.MON_SUMMARIES | L.MON_ACCEPT))
    job = O.OracleJob(**cases.oracle_kwargs(case, layout=eng.layout()), want_hist=True)
    eng.init_state_normal(); job.init_state_normal()
    eng.run(case["nsteps"]); job.run(case["nsteps"])
    npost = len(range(case["burnin"] + 1, case["nsteps"] + 1, case["thinning"]))
    for c in (0, 17, case["nchains"] - 1):
        v = eng.chain(c)
        assert v.shape == (100, npost) and v.flags.f_contiguous
        assert np.array_equal(v, job.hist[:, c, :].T)
    # mean(chain) of stats/mean.jl:7-11 from the stored history == from the running sums (same order)
    s, _, n = eng.chain_sums()
    assert np.allclose(eng.chain(3).mean(axis=1), s[3] / n, rtol=1e-13)
    eng.close()


@pytest.mark.parametrize("name", ["hmc_dense_d37", "mala_dense_d100", "hmc_d100", "hmc_rats", "hmc_dense_d192_stream_mean", "hmc_logitm_d33_n70", "mala_logitm_d20"])
def test_history_of_all_monitored_fields(name):
    case = cases.make_case(name)
    mon = L.MON_HISTORY | L.MON_HIST_LT | L.MON_HIST_GRAD | L.MON_SUMMARIES | L.MON_ACCEPT
    eng = K.Engine(**cases.engine_kwargs(case, monitor=mon, steps_per_launch=3))
    job = O.OracleJob(**cases.oracle_kwargs(case, layout=eng.layout()), want_hist=True)
    if case["x0"] is None:
        eng.init_state_normal(); job.init_state_normal()
    else:
        eng.set_state(case["x0"]); job.set_state(case["x0"])
    eng.run(case["nsteps"]); job.run(case["nsteps"])
    for c in (0, case["nchains"] // 2, case["nchains"] - 1):
        lt, g = eng.chain_fields(c, True, True)
        assert np.array_equal(eng.chain(c), job.hist[:, c, :].T)
        assert np.array_equal(lt, job.hist_lt[:, c]) and np.array_equal(g, job.hist_g[:, c, :].T)
    eng.close()


@pytest.mark.parametrize("spl,ring", [(7, 0), (0, 0), (5, 4)])
def test_slice_sampler_out_of_lockstep_keeps_history_sums_and_logtargets(spl, ring):
    """The free-running slice kernel (klara_diagt_slice.h: every lane takes its elements through a launch on its own) with every monitor it serves:
    value history (a machine stores its element of a saved state when ITS transition ends), log-target history (formed afterwards from the saved
    values, k_diagt_hist_lt), running sums, accept diagnostics — burn-in 4, thinning 3, several launches, a history ring — against the oracle
    bit for bit: final state, sums, every saved column of three chains (the ragged last wavefront group among them)."""
    case = dict(cases.make_case("slice_d20_stepout"), nchains=45, nsteps=31, burnin=4, thinning=3)
    case["x0"] = np.random.default_rng(12).standard_normal((45, 20))
    mon = L.MON_HISTORY | L.MON_HIST_LT | L.MON_SUMMARIES | L.MON_ACCEPT
    eng = K.Engine(**cases.engine_kwargs(case, monitor=mon, steps_per_launch=spl, hist_ring_cols=ring))
    assert eng.layout()[0] == 3
    job = O.OracleJob(**cases.oracle_kwargs(case, layout=eng.layout()), want_hist=True)
    eng.set_state(case["x0"]); job.set_state(case["x0"])
    for k in (13, 18):
        eng.run(k); assert job.run(k) == 0
    _assert_same(eng, job, case)
    nsaved = eng.saved_steps()
    assert nsaved == 9
    keep = min(ring, nsaved) if ring else nsaved
    for c in (0, 22, 44):
        lt, _ = eng.chain_fields(c, True, False)
        assert np.array_equal(eng.chain(c), job.hist[nsaved - keep:, c, :].T), c
        assert np.array_equal(lt, job.hist_lt[nsaved - keep:, c]), c
    eng.close()


def test_monitored_fields_and_iostream_sink(tmp_path):
    """doc/examples/swiss/MALA/analytical.jl:36-44: monitor [:value, :logtarget, :gradlogtarget], diagnostics
    [:accept] — kept per saved step on device, bit-equal to the oracle, and written by the :iostream destination
    in the reference's CSV layout."""
    case = cases.make_case("mala_swiss")
    X, y = cases.swiss_data()
    p = K.BasicContMuvParameter("p", logtarget=K.LogisticTarget(X, y, 100.0))
    job = K.BasicMCJob(K.likelihood_model(p, False), K.MALA(0.1), K.BasicMCRange(nsteps=40, burnin=10), {"p": case["x0"]}, seed=20260927,
                       outopts={"destination": "iostream", "filepath": str(tmp_path), "monitor": ["value", "logtarget", "gradlogtarget"],
                                "diagnostics": ["accept"]})
    o = O.OracleJob(**cases.oracle_kwargs(case, layout=job.engine.layout()), want_hist=True)
    o.set_state(case["x0"]); o.run(40)
    K.run(job)
    chain = K.output(job)
    for c in (0, 41, 69):
        assert np.array_equal(chain.value(c), o.hist[:, c, :].T)
        assert np.array_equal(chain.logtarget(c), o.hist_lt[:, c])
        assert np.array_equal(chain.gradlogtarget(c), o.hist_g[:, c, :].T)
    d = tmp_path / "chain_42"
    from klara_jl_amd.iostream import julia_float_repr as j
    lines = (d / "value.csv").read_text().splitlines()
    assert len(lines) == 30 and lines[0] == ",".join(j(v) for v in o.hist[0, 41])
    assert (d / "logtarget.csv").read_text().splitlines()[29] == j(o.hist_lt[29, 41])
    assert (d / "gradlogtarget.csv").read_text().splitlines()[3] == ",".join(j(v) for v in o.hist_g[3, 41])
    assert (d / "diagnosticvalues.csv").read_text().splitlines() == ["true" if a else "false" for a in o.accept[10:, 41]]
    # read(iostream, Float64) (BasicContParamIOStream.jl:254-281): the files read back as the chain, bit for bit
    back = K.read_chain(str(d))
    assert back.size == 4 and back.n == 30 and np.array_equal(back.value, o.hist[:, 41, :].T) and np.array_equal(back.logtarget, o.hist_lt[:, 41])
    assert np.array_equal(back.gradlogtarget, o.hist_g[:, 41, :].T) and np.array_equal(back.diagnosticvalues[0], o.accept[10:, 41].astype(bool))
    assert np.allclose(K.acceptance(chain), o.accept[10:].mean(axis=0))
    job.close()


def test_device_mcvar_matches_host_estimators():
    """stats/variance/mcvar.jl (:iid, :bm, :imse), convergence/ess.jl, iact.jl on device for all chains at once vs
    the NumPy restatement (klara_jl_amd.stats) on individual chains; tolerance 1e-9 relative (different summation)."""
    p = K.BasicContMuvParameter("p", logtarget=K.GaussDiagTarget.negdot(3))
    job = K.BasicMCJob(K.likelihood_model(p, False), K.MH(np.full(3, 0.6)), K.BasicMCRange(nsteps=2600, burnin=100),
                       {"p": np.zeros((70, 3))}, seed=20260927)
    K.run(job)
    chain = K.output(job)
    iid, bm, imse = (K.chain_mcvar(chain, t, batchlen=50) for t in ("iid", "bm", "imse"))
    ess, iact = K.chain_ess(chain), K.chain_iact(chain)
    assert iid.shape == (70, 3)
    for c in (0, 33, 69):
        v = chain.value(c)
        assert np.allclose(iid[c], K.stats.mcvar_chain(v, "iid"), rtol=1e-9)
        assert np.allclose(bm[c], K.stats.mcvar_chain(v, "bm", 50), rtol=1e-9)
        assert np.allclose(imse[c], K.stats.mcvar_chain(v, "imse"), rtol=1e-7)
        assert np.allclose(ess[c], K.stats.ess_chain(v), rtol=1e-7) and np.allclose(iact[c], K.stats.iact_chain(v), rtol=1e-7)
    assert np.all(iact > 1.0) and np.all(ess < 2500)          # a random-walk chain is autocorrelated
    job.close()


# ------------------------------------------------------------------ reference-API level
def test_readme_flow_basic_mc_job():
    """README.md:23-66 through the host mirror: MH on lt = -dot(z,z), 10000 steps, burnin 1000."""
    p = K.BasicContMuvParameter("p", logtarget=K.GaussDiagTarget.negdot(2))
    model = K.likelihood_model(p, False)
    job = K.BasicMCJob(model, K.MH(np.ones(2)), K.BasicMCRange(nsteps=10000, burnin=1000),
                       {"p": np.tile([5.1, -0.9], (256, 1))}, outopts={"diagnostics": ["accept"]}, seed=20260927)
    K.run(job)
    chain = K.output(job)
    m = K.mean(chain)                         # (nchains x D)
    assert chain.value(0).shape == (2, 9000)
    assert np.allclose(K.mean(chain, 0), chain.value(0).mean(axis=1), rtol=1e-12)
    assert abs(m.mean()) < 0.01                                 # truth 0
    s, q, n = chain._sums
    assert np.all(np.abs(q.sum(0) / (n * 256) - 0.5) < 0.01)    # truth var 1/2
    acc = K.acceptance(chain)
    assert acc.shape == (256,) and 0.35 < acc.mean() < 0.5
    # one replica against the oracle bit for bit (BASELINE cfg 1 is the reference's CPU-runnable case)
    o = O.OracleJob(sampler=L.SAMPLER_MH, target_kind=L.TARGET_GAUSS_DIAG, nchains=1, ndims=2, nsteps=10000,
                    burnin=1000, mh_sigma=[1.0, 1.0], layout=job.engine.layout())
    o.set_state([[5.1, -0.9]]); o.run(10000)
    assert np.array_equal(o.sum[0], s[0]) and np.array_equal(o.accept[:, 0], job.engine.accept_mask()[:, 0])
    job.close()


# ------------------------------------------------------------------ randomized configurations
def _random_case(seed, wide=False, logit_mfma=False, split=False):
    """One job drawn from the whole configuration space the library accepts: target family and size, sampler, tuner,
    range, chain count (valid combinations only — the refused ones are in test_error_paths).  wide: the sizes round 4 moved onto hand-written
    kernels — dense targets of 129..256 dimensions (streamed matrix-core layouts) and logistic regressions with 9..16 parameters (row split)."""
    rng = np.random.default_rng((13000 if split else 9000 if logit_mfma else 5000 if wide else 1000) + seed)
    fam = "dense_split" if split else "logitm" if logit_mfma else rng.choice(["dense", "logit"]) if wide else rng.choice(["diag_unit", "diag", "dense", "logit", "hier", "custom"], p=[0.2, 0.25, 0.15, 0.13, 0.14, 0.13])
    if fam == "logitm":       # round 6: 17 .. 128 parameters on the matrix cores (klara_logit_mfma.h): every NE, rows that end inside a tile / a block of tiles
        d = int(rng.choice([17, 20, 31, 32, 33, 48, 64, 65, 96, 97, 128, 129, 192, 256]))
        n = int(rng.choice([1, 15, 16, 17, 32, 33, 100, 300]))
        X, y = cases.synthetic_logit(n, d, seed=seed)
        target = K.LogisticTarget(X / np.sqrt(d), y, float(rng.choice([1.0, 100.0])))
        fam = "logit"
    elif fam == "dense_split":    # round 6: dense targets of 257 .. 1,024 dimensions on the workgroup-split layout (8 / 12 / 16 wavefronts per tile, every tile deal)
        d = int(rng.choice([257, 272, 289, 320, 333, 384, 385, 448, 512, 513, 576, 641, 700, 768, 769, 900, 1000, 1024]))
        target = K.GaussDenseTarget(cases.compound_symmetric_precision(d, float(rng.uniform(0.1, 0.7))), const=float(rng.uniform(-2, 2)),
                                    mu=(rng.uniform(-1.5, 1.5, d) if rng.integers(0, 2) else None))
        fam = "dense"
    elif fam == "diag_unit":
        d = int(rng.choice([1, 2, 3, 5, 8, 16, 17, 19, 31, 40, 63, 77, 100, 128, 129, 200, 300]))
        target = K.GaussDiagTarget.negdot(d)
    elif fam == "diag":
        d = int(rng.choice([2, 4, 7, 16, 18, 23, 48, 65, 96, 127, 140]))
        target = K.GaussDiagTarget.mvnormal(rng.uniform(-2, 2, d), rng.uniform(0.5, 2.0, d))
    elif fam == "dense":
        d = int(rng.choice([129, 144, 160, 161, 177, 192, 193, 224, 225, 256])) if wide else int(rng.choice([3, 8, 20, 33, 50, 70]))
        target = K.GaussDenseTarget(cases.compound_symmetric_precision(d, float(rng.uniform(0.1, 0.7))), const=float(rng.uniform(-2, 2)),
                                    mu=(rng.uniform(-1.5, 1.5, d) if rng.integers(0, 2) else None))
    elif fam == "logit":
        d = int(rng.choice([9, 10, 12, 13, 15, 16])) if wide else int(rng.choice([1, 2, 3, 4, 6, 8, 11]))
        n = int(rng.choice([5, 30, 63, 64, 100, 300, 1100 if wide else 300]))   # (11: 9..16 parameters run on the row-split kernels since round 4; 1,100 rows x 17 columns do not fit the LDS: closure form)
        X, y = cases.synthetic_logit(n, d, seed=seed)
        target = K.LogisticTarget(X, y, float(rng.choice([1.0, 100.0])))
    elif fam == "hier":
        R = int(rng.choice([2, 5, 8, 9, 12, 21, 32, 40])); T = int(rng.integers(2, 8))
        xc = np.linspace(-10.0, 10.0, T) + float(rng.uniform(-1, 1))
        a = 200.0 + 10.0 * rng.standard_normal(R); b = 5.0 + 0.5 * rng.standard_normal(R)
        target = K.HierNormalTarget(a[:, None] + b[:, None] * xc[None, :] + 5.0 * rng.standard_normal((R, T)), xc)
        d = target.ndims
    else:
        d = int(rng.choice([2, 5, 9, 16, 24]))
        target = K.CustomTarget(d, cases.SRC_QUARTIC_CHAIN, [float(rng.uniform(0.01, 0.2)), float(rng.uniform(0.1, 0.8))])
    samplers = [L.SAMPLER_MH, L.SAMPLER_MALA, L.SAMPLER_HMC] + ([] if (fam == "dense" and d > 70) or (logit_mfma and d > 48) else [L.SAMPLER_SLICE])   # (dense slice: D full evaluations per probe on the oracle's side)
    sampler = int(rng.choice(samplers))
    scale = 0.02 if fam == "hier" else (0.5 if logit_mfma else 0.05 if fam == "logit" else (0.06 if split else 0.1 if wide else 0.3))
    c = dict(sampler=sampler, target=target, nchains=int(rng.choice([1, 2, 7, 8, 9, 33, 64, 100, 131] if not split else [1, 7, 16, 17, 33, 50])), x0=None, seed=int(rng.integers(1, 2 ** 40)),
             name=f"random_{seed}_{fam}")
    if sampler == L.SAMPLER_MH:
        c["mh_sigma"] = np.full(d, scale) * rng.uniform(0.5, 1.5, d)
    elif sampler == L.SAMPLER_MALA:
        c["driftstep"] = float(scale * scale * rng.uniform(0.5, 2.0))
    elif sampler == L.SAMPLER_HMC:
        c["leapstep"] = float(scale * rng.uniform(0.3, 0.8)); c["nleaps"] = int(rng.integers(1, 7))
    else:
        c["slice_widths"] = np.full(d, 4 * scale) * rng.uniform(0.5, 1.5, d); c["slice_stepout"] = bool(rng.integers(0, 2))
    c["burnin"] = int(rng.choice([0, 3, 20])); c["thinning"] = int(rng.choice([1, 1, 2, 5]))
    c["nsteps"] = c["burnin"] + int(rng.integers(6, 40))
    if split:                          # (the oracle's side: D^2 flop per gradient on one core per chain)
        c["burnin"] = min(c["burnin"], 3); c["nsteps"] = c["burnin"] + int(rng.integers(4, 14))
    tun = rng.choice(["vanilla", "verbose", "rate", "rate_erf", "pooled", "da"])
    if tun == "verbose":
        c.update(verbose=True, period=int(rng.integers(3, 12)))
    elif tun in ("rate", "rate_erf", "pooled") and not (tun == "pooled" and sampler == L.SAMPLER_SLICE):
        c.update(tuner=L.TUNER_ACCEPT_RATE, targetrate=float(rng.uniform(0.3, 0.8)), period=int(rng.integers(3, 12)))
        if tun == "rate_erf":
            c.update(tuner_score=1, score_k=3.0)
        if tun == "pooled":
            c.update(tuner_mode=L.TUNE_POOLED)
    elif tun == "da" and sampler == L.SAMPLER_HMC:
        c.update(tuner=L.TUNER_DUAL_AVERAGING, targetrate=float(rng.uniform(0.5, 0.8)), da_nadapt=int(rng.integers(5, 30)))
    if fam in ("hier", "logit"):
        if fam == "hier":
            c["x0"] = target.least_squares_start()[None, :] + 0.02 * rng.standard_normal((c["nchains"], d))
        else:
            c["x0"] = 0.1 * rng.standard_normal((c["nchains"], d))
    return c, rng


_BIG = [(smp, d, mu) for smp in ("mh", "mala", "hmc", "hmc_rate", "hmc_da") for d in (130, 161, 193, 256) for mu in (False, True)] + \
       [(smp, d, mu) for smp in ("mh", "mala", "hmc", "hmc_da") for d in (21, 37, 70, 128) for mu in (False, True)]      # (P in LDS, klara_dense.h: NE = 8, 16, 25, 32)
_BIG += [("slice", d, mu) for d in (130, 161, 193, 256) for mu in (False, True)]       # (round 5: the slice sampler on the streamed layouts; step-out on without a mean, off with one)


@pytest.mark.parametrize("smp,d,mu", _BIG, ids=[f"{a}-d{b}-{'mean' if m else 'nomean'}" for a, b, m in _BIG])
def test_every_streamed_dense_instantiation_in_one_launch(smp, d, mu):
    """Every k_dense_big<sampler, NE, mean, dual averaging> instantiation (NE = 40 / 48 / 56 / 64), and the LDS-resident kernels below 129 dimensions at four
    sizes, with ALL transitions of the job in ONE launch, a ragged
    second tile, running sums and histories on, at step sizes where a good share of the proposals is rejected and a good share accepted: the state a
    lane carries from one transition to the next — kept after an accept, re-read after a reject — is what this pins.  (k_dense_big<MH, 48, mean> once lost
    an element of every chain that had just rejected: a register copy the compiler placed under a divergent branch's execution mask; a launch of one
    transition cannot see that, and the randomised jobs only run into it by the luck of their launch sizes.)"""
    rng = np.random.default_rng(d + 7 * len(smp) + int(mu))
    target = K.GaussDenseTarget(cases.compound_symmetric_precision(d, 0.4), const=0.3, mu=(rng.uniform(-1.5, 1.5, d) if mu else None))
    c = dict(target=target, nchains=21, x0=None, seed=4242 + d, name=f"big_{smp}_{d}_{int(mu)}", burnin=2, thinning=2, nsteps=14)
    if smp == "slice":               # (a probe is a full evaluation: D x ~6 of them per transition and chain on the oracle's side)
        c.update(sampler=L.SAMPLER_SLICE, slice_widths=np.linspace(0.5, 2.0, d), slice_stepout=not mu, nsteps=5, burnin=1)
    elif smp == "mh":
        c.update(sampler=L.SAMPLER_MH, mh_sigma=np.full(d, 0.07) * rng.uniform(0.7, 1.3, d))
    elif smp == "mala":
        c.update(sampler=L.SAMPLER_MALA, driftstep=0.3)
    else:
        c.update(sampler=L.SAMPLER_HMC, leapstep=0.45, nleaps=3)
        if smp == "hmc_rate":
            c.update(tuner=L.TUNER_ACCEPT_RATE, targetrate=0.6, period=4)
        if smp == "hmc_da":
            c.update(tuner=L.TUNER_DUAL_AVERAGING, targetrate=0.65, da_nadapt=9)
    nograd = smp in ("mh", "slice")
    mon = L.MON_ACCEPT | L.MON_SUMMARIES | L.MON_HISTORY | L.MON_HIST_LT | (0 if nograd else L.MON_HIST_GRAD)
    eng = K.Engine(**cases.engine_kwargs(c, monitor=mon, steps_per_launch=0, nstreams=1))
    assert eng.layout() == (1, 4, 8 * ((d + 31) // 32) if d > 128 else {21: 8, 37: 16, 70: 25, 128: 32}[d]), eng.layout()
    job = O.OracleJob(**cases.oracle_kwargs(c, layout=eng.layout()), want_hist=True)
    eng.init_state_normal(); assert job.init_state_normal() == 0
    eng.run(c["nsteps"]); assert job.run(c["nsteps"]) == 0
    rate = job.accept.mean()
    assert rate == 1.0 if smp == "slice" else 0.05 < rate < 0.95, rate                    # both the commit and the re-read are exercised
    _assert_same(eng, job, c)
    for ch in (0, 17, 20):
        assert np.array_equal(eng.chain(ch), job.hist[:, ch, :].T), "history differs"
        lt, g = eng.chain_fields(ch, logtarget=True, gradlogtarget=not nograd)
        assert np.array_equal(lt, job.hist_lt[:, ch])
        if not nograd:
            assert np.array_equal(g, job.hist_g[:, ch, :].T)
    eng.close()


_SPLIT = [(smp, d, mu) for smp in ("mh", "mala", "mala_pooled", "hmc", "hmc_rate", "hmc_da") for d, mu in ((270, False), (330, True), (520, False), (780, True), (1010, False))]
_SPLIT += [("slice", 270, False), ("slice", 330, True), ("slice", 780, True)]      # (a probe is a full evaluation: D x ~6 per transition and chain on the oracle's side)


@pytest.mark.parametrize("smp,d,mu", _SPLIT, ids=[f"{a}-d{b}-{'mean' if m else 'nomean'}" for a, b, m in _SPLIT])
def test_every_split_dense_instantiation_in_one_launch(smp, d, mu):
    """Every k_dense_split<sampler, dual averaging, mean, registers> instantiation (round 6: 8 / 12 / 16 wavefronts per tile, 2 .. 4 row tiles each, an odd first tile
    in some of them), with ALL transitions of the job in ONE launch, a ragged second tile, running sums and histories on, at step sizes where a good share of the
    proposals is rejected and a good share accepted: what a wavefront carries from one transition to the next — the proposal in its LDS column, the committed state
    in X / GR — and every barrier between the wavefronts of a tile is what this pins."""
    rng = np.random.default_rng(d + 7 * len(smp) + int(mu))
    target = K.GaussDenseTarget(cases.compound_symmetric_precision(d, 0.4), const=0.3, mu=(rng.uniform(-1.5, 1.5, d) if mu else None))
    c = dict(target=target, nchains=21, x0=None, seed=4242 + d, name=f"split_{smp}_{d}_{int(mu)}", burnin=2, thinning=2, nsteps=14)
    sc = 256.0 / d
    if smp == "slice":
        c.update(sampler=L.SAMPLER_SLICE, slice_widths=np.linspace(0.5, 2.0, d), slice_stepout=not mu, nsteps=3, burnin=1, nchains=18 if d > 400 else 21)
    elif smp == "mh":
        c.update(sampler=L.SAMPLER_MH, mh_sigma=np.full(d, 0.07 * sc ** 0.5) * rng.uniform(0.7, 1.3, d))
    elif smp in ("mala", "mala_pooled"):
        c.update(sampler=L.SAMPLER_MALA, driftstep=0.3 * sc ** (1.0 / 3.0))
        if smp == "mala_pooled":
            c.update(tuner=L.TUNER_ACCEPT_RATE, tuner_mode=L.TUNE_POOLED, targetrate=0.574, period=4)
    else:
        c.update(sampler=L.SAMPLER_HMC, leapstep=0.45 * sc ** 0.25, nleaps=3)
        if smp == "hmc_rate":
            c.update(tuner=L.TUNER_ACCEPT_RATE, targetrate=0.6, period=4)
        if smp == "hmc_da":
            c.update(tuner=L.TUNER_DUAL_AVERAGING, targetrate=0.65, da_nadapt=9)
    nograd = smp in ("mh", "slice")
    mon = L.MON_ACCEPT | L.MON_SUMMARIES | L.MON_HISTORY | L.MON_HIST_LT | (0 if nograd else L.MON_HIST_GRAD)
    eng = K.Engine(**cases.engine_kwargs(c, monitor=mon, steps_per_launch=0, nstreams=1))
    assert eng.layout() == O.split_dense_layout(d), eng.layout()
    job = O.OracleJob(**cases.oracle_kwargs(c, layout=eng.layout()), want_hist=True)
    eng.init_state_normal(); assert job.init_state_normal() == 0
    eng.run(c["nsteps"]); assert job.run(c["nsteps"]) == 0
    rate = job.accept.mean()
    assert rate == 1.0 if smp == "slice" else 0.05 < rate < 0.95, rate                    # both the commit and the re-read are exercised
    _assert_same(eng, job, c)
    for ch in (0, 17, c["nchains"] - 1):
        assert np.array_equal(eng.chain(ch), job.hist[:, ch, :].T), "history differs"
        lt, g = eng.chain_fields(ch, logtarget=True, gradlogtarget=not nograd)
        assert np.array_equal(lt, job.hist_lt[:, ch])
        if not nograd:
            assert np.array_equal(g, job.hist_g[:, ch, :].T)
    eng.close()


@pytest.mark.parametrize("seed", range(48))
def test_random_configurations_wide(seed):
    """48 more jobs from the sizes that moved onto hand-written kernels in round 4 (dense 129..256 dimensions on the streamed matrix-core layouts — HMC with
    every tuner, MALA, MH; logistic regression with 9..16 parameters), run like the others."""
    _run_random(*_random_case(seed, wide=True))


@pytest.mark.parametrize("seed", range(32))
def test_random_configurations_dense_on_the_split_layout(seed):
    """32 dense targets of 257 .. 1,024 dimensions (layout kind 6, klara_dense_split.h; round 6): MH / MALA / HMC, every tuner, any range and chain count, random launch
    splitting, a shard's chain offset — run like the others."""
    c, rng = _random_case(seed, split=True)
    e = K.Engine(**cases.engine_kwargs(c))
    assert e.layout()[0] == 6
    e.close()
    _run_random(c, rng)


@pytest.mark.parametrize("seed", range(40))
def test_random_configurations_logistic_on_the_matrix_cores(seed):
    """40 logistic regressions with 17 .. 128 parameters (layout kind 5, klara_logit_mfma.h; round 6): MH / MALA / HMC, every tuner, any range and chain count, random
    launch splitting — run like the others."""
    c, rng = _random_case(seed, logit_mfma=True)
    e = K.Engine(**cases.engine_kwargs(c))
    assert e.layout()[0] == 5
    e.close()
    _run_random(c, rng)


_LOGITM = [(smp, d, n) for smp in ("mh", "mala", "mala_pooled", "hmc", "hmc_rate", "hmc_da", "slice") for d, n in ((17, 50), (40, 16), (96, 33), (128, 70), (160, 20), (256, 33)) if not (smp == "slice" and d > 128)]


@pytest.mark.parametrize("smp,d,n", _LOGITM, ids=[f"{a}-d{b}-n{c_}" for a, b, c_ in _LOGITM])
def test_every_matrix_core_logistic_instantiation_in_one_launch(smp, d, n):
    """Every k_logit_mfma<sampler, NE, dual averaging> instantiation (NE = 8 / 16 / 24 / 32) with ALL transitions of the job in ONE launch, a ragged second tile of
    chains, data rows that end inside a tile (or fill one exactly), running sums and histories on, at steps where a good share of the proposals is rejected and a good
    share accepted: the state a lane carries from one transition to the next — kept after an accept, re-read after a reject — the fragment ring across the row blocks
    of consecutive evaluations, and the padding rows' masks."""
    rng = np.random.default_rng(d + 7 * len(smp) + n)
    X, y = cases.synthetic_logit(n, d, seed=d + n)
    c = dict(target=K.LogisticTarget(X / np.sqrt(d), y, 10.0), nchains=21, x0=0.3 * rng.standard_normal((21, d)), seed=777 + d, name=f"logitm_{smp}_{d}_{n}",
             burnin=2, thinning=2, nsteps=14)
    if smp == "slice":               # (a probe is a full evaluation: D x ~6 of them per transition and chain on the oracle's side)
        c.update(sampler=L.SAMPLER_SLICE, slice_widths=np.linspace(0.8, 3.0, d), slice_stepout=d != 96, nsteps=4, burnin=1)
    elif smp == "mh":
        c.update(sampler=L.SAMPLER_MH, mh_sigma=np.full(d, 1.2 / np.sqrt(d)) * rng.uniform(0.7, 1.3, d))
    elif smp.startswith("mala"):
        c.update(sampler=L.SAMPLER_MALA, driftstep=3.0 / d ** (1 / 3))
        if smp == "mala_pooled":
            c.update(tuner=L.TUNER_ACCEPT_RATE, tuner_mode=L.TUNE_POOLED, targetrate=0.574, period=4)
    else:
        c.update(sampler=L.SAMPLER_HMC, leapstep=1.15, nleaps=3)
        if smp == "hmc_rate":
            c.update(tuner=L.TUNER_ACCEPT_RATE, targetrate=0.6, period=4)
        if smp == "hmc_da":
            c.update(tuner=L.TUNER_DUAL_AVERAGING, targetrate=0.65, da_nadapt=9)
    nograd = smp in ("mh", "slice")
    mon = L.MON_ACCEPT | L.MON_SUMMARIES | L.MON_HISTORY | L.MON_HIST_LT | (0 if nograd else L.MON_HIST_GRAD)
    eng = K.Engine(**cases.engine_kwargs(c, monitor=mon, steps_per_launch=0, nstreams=1))
    assert eng.layout() == (5, 4, 8 * ((d + 31) // 32)), eng.layout()
    job = O.OracleJob(**cases.oracle_kwargs(c, layout=eng.layout()), want_hist=True)
    eng.set_state(c["x0"]); assert job.set_state(c["x0"]) == 0
    eng.run(c["nsteps"]); assert job.run(c["nsteps"]) == 0
    rate = job.accept.mean()
    assert rate == 1.0 if smp == "slice" else 0.05 < rate < 0.95, rate                       # both the commit and the re-read are exercised
    _assert_same(eng, job, c)
    for ch in (0, 17, 20):
        assert np.array_equal(eng.chain(ch), job.hist[:, ch, :].T), "history differs"
        lt, g = eng.chain_fields(ch, logtarget=True, gradlogtarget=not nograd)
        assert np.array_equal(lt, job.hist_lt[:, ch])
        if not nograd:
            assert np.array_equal(g, job.hist_g[:, ch, :].T)
    eng.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("KLARA_RANDOM_FIRST", "0")), int(os.environ.get("KLARA_RANDOM_FIRST", "0")) + int(os.environ.get("KLARA_RANDOM_CASES", "96"))))
def test_random_configurations(seed):
    """96 jobs (KLARA_RANDOM_CASES overrides the count, KLARA_RANDOM_FIRST the first seed) drawn at random from the accepted configuration space, run in randomly sized pieces with a random number of
    transitions per launch and every monitor on: accept mask, state, sums, tuner state and one chain's full history must
    equal the oracle's bit for bit."""
    _run_random(*_random_case(seed))


def _run_random(c, rng):
    mon = L.MON_ACCEPT | L.MON_SUMMARIES | L.MON_HISTORY | L.MON_HIST_LT
    if c["sampler"] in (L.SAMPLER_MALA, L.SAMPLER_HMC):
        mon |= L.MON_HIST_GRAD
    off = int(rng.choice([0, 0, 5, (1 << 33) + 12345]))          # this shard's first global chain id (Philox subsequence)
    eng = K.Engine(**cases.engine_kwargs(c, monitor=mon, steps_per_launch=int(rng.choice([0, 1, 2, 5, 16])),
                                         nstreams=int(rng.choice([0, 1, 3])), chain_offset=off))
    job = O.OracleJob(**cases.oracle_kwargs(c, layout=eng.layout(), chain_offset=off), want_hist=True)
    if c["x0"] is None:
        eng.init_state_normal(); assert job.init_state_normal() == 0
    else:
        eng.set_state(c["x0"]); assert job.set_state(c["x0"]) == 0
    left = c["nsteps"]
    while left > 0:
        k = int(rng.integers(1, left + 1)); left -= k
        ost = job.run(k)
        try:
            eng.run(k); gst = 0
        except K.KlaraError as e:
            gst = e.status
        assert gst == ost, (gst, ost)
        if ost != 0:                          # a slice that shrank onto the current point (SliceSampler.jl:102): both sides say so
            assert ost == L.ERR_SLICE_STUCK
            eng.close()
            return
    _assert_same(eng, job, c)
    ch = int(rng.integers(0, c["nchains"]))
    assert np.array_equal(eng.chain(ch), job.hist[:, ch, :].T), "history differs"
    lt, g = eng.chain_fields(ch, logtarget=True, gradlogtarget=bool(mon & L.MON_HIST_GRAD))
    assert np.array_equal(lt, job.hist_lt[:, ch])
    if mon & L.MON_HIST_GRAD:
        assert np.array_equal(g, job.hist_g[:, ch, :].T)
    eng.close()


def test_dual_averaging_padding_lanes_do_not_set_the_trip_count():
    """A ragged last wavefront carries padding lanes whose phantom chain (x = 0, re-read every transition) can drive the dual-averaging
    step towards 0 when 0 is far out in the target's tail — nleaps = round(lambda / step) (iterate/HMC.jl:142-144) then reaches the cap
    of 65,536 and the wavefront, which runs to its longest trajectory, spends seconds per transition (30 s for this 30-transition job
    before padding lanes were pinned to one leapfrog)."""
    for name in ("hmc_dense_d70_mean_dualavg", "hmc_d40_dualavg"):
        c = dict(cases.make_case(name))
        if name == "hmc_d40_dualavg":          # the same on the pair-transposed layout: a mean far from the padding lanes' x = 0
            c["target"] = K.GaussDiagTarget.mvnormal(np.full(40, 30.0), np.linspace(0.6, 1.6, 40)); c["nchains"] = 27
            c["x0"] = 30.0 + np.random.default_rng(2).standard_normal((27, 40))
        eng = K.Engine(**cases.engine_kwargs(c))
        eng.set_state(c["x0"])
        eng.run(c["nsteps"])
        assert eng.last_run_ms()[0] < 2000.0, (name, eng.last_run_ms())
        eng.close()


# ------------------------------------------------------------------ streaming batch means
@pytest.mark.parametrize("name,batchlen,streams", [("mala_d100", 7, 0), ("dt_hmc_d100", 5, 2), ("hmc_dense_d37", 6, 0),
                                                    ("mala_swiss", 9, 0), ("hmc_rats", 4, 0), ("slice_d5", 3, 0),
                                                    ("custom_banana_hmc", 8, 0), ("mala_dense_d512_split_mean_tuned", 5, 0), ("pair_quartic_slice_d100", 2, 0)])
def test_streaming_batch_means(name, batchlen, streams):
    """klara_desc.bm_batchlen: mcvar(:bm) (mcvar.jl:35-41) without stored history.  Bit-exact against the oracle closing the
    same batches from the same running sums, and equal (to rounding) to the post-hoc estimator over the stored history —
    on every layout kind, across launch splitting, thinning and chain partitions on two streams."""
    c = dict(cases.make_case(name)); c.update(nsteps=max(c["nsteps"], 120), thinning=2, burnin=11)
    if c.get("da_nadapt"):
        c["da_nadapt"] = 40
    eng = K.Engine(**cases.engine_kwargs(c, monitor=L.MON_SUMMARIES | L.MON_HISTORY, bm_batchlen=batchlen, steps_per_launch=7,
                                         nstreams=streams))
    eng.set_state(c["x0"]) if c.get("x0") is not None else eng.init_state_normal()
    xstart = eng.state()[0]
    eng.run(50); eng.run(c["nsteps"] - 50)                 # a batch boundary inside each call and across the two
    bm, nb = eng.chain_bm()
    nsaved = (c["nsteps"] - 11 - 1) // 2 + 1
    assert nb == nsaved // batchlen and nb >= 2
    job = O.OracleJob(**cases.oracle_kwargs(c, layout=eng.layout()))
    job.set_state(c["x0"]) if c.get("x0") is not None else job.init_state_normal()
    obm, onb = job.run_with_batch_means(c["nsteps"], batchlen)
    assert onb == nb and np.array_equal(bm, obm)
    posthoc = eng.chain_mcvar(batchlen, 0)[1]              # k_chain_stats over the history: first nb*batchlen samples, two-pass
    assert np.allclose(bm, posthoc, rtol=1e-8, atol=1e-24)   # (a chain that never moved: 0 against rounding noise of 1e-32)
    # set_state clears the accumulators and keeps the stream (a replay); other launch boundaries, same batches
    eng.set_state(xstart); eng.run(c["nsteps"])
    bm2, nb2 = eng.chain_bm()
    assert nb2 == nb and np.array_equal(bm2, bm)
    eng.close()


def test_streaming_batch_means_through_the_job_api():
    """mcvar(chain, :bm) for 4,096 chains with destination none (nothing stored): the Monte Carlo standard error of the
    chain means predicts the spread of those means across independent chains."""
    p = K.BasicContMuvParameter("p", logtarget=K.GaussDiagTarget.negdot(3))
    job = K.BasicMCJob(K.likelihood_model(p, False), K.MALA(0.9), K.BasicMCRange(nsteps=6000, burnin=1000),
                       {"p": np.zeros((4096, 3))}, outopts={"destination": "none"}, bm_batchlen=100, seed=20260927)
    K.run(job)
    chain = K.output(job)
    v = K.chain_mcvar(chain, "bm", 100)                    # (chains, D): variance of each chain's mean
    means = K.mean(chain)
    assert v.shape == (4096, 3) and job.engine.chain_bm()[1] == 50
    ratio = means.var(axis=0) / v.mean(axis=0)             # batch means slightly underestimate at finite batch length
    assert np.all((0.9 < ratio) & (ratio < 1.25)), ratio
    job.close()


def test_streaming_batch_means_errors():
    with pytest.raises(K.KlaraError) as ei:                # needs the running sums
        K.Engine(sampler=L.SAMPLER_MH, target=K.GaussDiagTarget.negdot(2), nchains=4, nsteps=50, mh_sigma=[1.0, 1.0], bm_batchlen=5)
    assert ei.value.status == L.ERR_INVALID_ARG
    eng = K.Engine(sampler=L.SAMPLER_MH, target=K.GaussDiagTarget.negdot(2), nchains=4, nsteps=50, mh_sigma=[1.0, 1.0],
                   monitor=L.MON_SUMMARIES)
    eng.set_state(np.zeros((4, 2))); eng.run(10)
    with pytest.raises(K.KlaraError) as ei:
        eng.chain_bm()
    assert ei.value.status == L.ERR_STATE
    eng.close()
    eng = K.Engine(sampler=L.SAMPLER_MH, target=K.GaussDiagTarget.negdot(2), nchains=4, nsteps=50, mh_sigma=[1.0, 1.0],
                   monitor=L.MON_SUMMARIES, bm_batchlen=30)
    eng.set_state(np.zeros((4, 2))); eng.run(50)
    bm, nb = eng.chain_bm()
    assert nb == 1 and np.isnan(bm).all()                  # "Choose batch size such that the number of batches is > 1"
    eng.close()


# ------------------------------------------------------------------ user-defined targets (KLARA_TARGET_CUSTOM)
def _run_engine(case, **over):
    eng = K.Engine(**cases.engine_kwargs(case, monitor=L.MON_ACCEPT | L.MON_SUMMARIES, **over))
    eng.set_state(case["x0"]) if case.get("x0") is not None else eng.init_state_normal()
    eng.run(case["nsteps"])
    out = (eng.state(), eng.accept_mask(), eng.chain_sums()[:2], eng.layout())
    eng.close()
    return out


def test_custom_target_equals_builtin_families():
    """The README closure (-dot(z,z), -2z) and the swiss example's closures, handed over as user source and compiled at
    klara_create, give bit for bit what the built-in diagonal / logistic families give (same lane layout: one chain per
    lane, so even the summation order is the same)."""
    c = cases.make_case("custom_negdot_mala_d3")
    b = dict(c, target=K.GaussDiagTarget.negdot(3))
    X, y = cases.synthetic_logit(40, 4)
    c2 = cases.make_case("custom_logit_mala_d4")
    b2 = dict(c2, target=K.LogisticTarget(X, y, 100.0))
    for cu, bu in ((c, b), (c2, b2)):
        (sc, mc, qc, lc), (sb, mb, qb, lb) = _run_engine(cu), _run_engine(bu)
        assert lc == lb and lc[0] == 0 and lc[1] == 1
        assert all(np.array_equal(u, v) for u, v in zip(sc, sb))
        assert np.array_equal(mc, mb) and 0 < mc.sum() < mc.size
        assert all(np.array_equal(u, v) for u, v in zip(qc, qb))


def test_custom_target_compile_error_and_limits():
    with pytest.raises(K.KlaraError) as ei:
        K.Engine(sampler=L.SAMPLER_MH, target=K.CustomTarget(2, "double klara_user_logtarget(const double* x) { return x[0] }"),
                 nchains=4, nsteps=5, mh_sigma=[1.0, 1.0])
    assert ei.value.status == L.ERR_COMPILE and "klara_user_target:1" in ei.value.log
    with pytest.raises(K.KlaraError) as ei:               # MALA needs the gradient closure
        K.Engine(sampler=L.SAMPLER_MALA, target=K.CustomTarget(2, cases.SRC_BANANA_LT_ONLY), nchains=4, nsteps=5, driftstep=0.1)
    assert ei.value.status == L.ERR_COMPILE and "klara_user_gradlogtarget" in ei.value.log
    with pytest.raises(K.KlaraError) as ei:               # D <= 256: the whole vector lives in one lane
        K.Engine(sampler=L.SAMPLER_MH, target=K.CustomTarget(1025, cases.SRC_NEGDOT), nchains=4, nsteps=5, mh_sigma=np.ones(1025))
    assert ei.value.status == L.ERR_UNSUPPORTED
    # a closure that is not finite at the start: the reference's initialize! assert (MH.jl:83)
    src = "KLARA_USER_FN double klara_user_logtarget(const double* x, int D, const double* data, long long ndata) { return kd_log(x[0]); }"
    eng = K.Engine(sampler=L.SAMPLER_MH, target=K.CustomTarget(1, src), nchains=4, nsteps=5, mh_sigma=[1.0])
    with pytest.raises(K.KlaraError) as ei:
        eng.set_state(-np.ones((4, 1)))
    assert ei.value.status == L.ERR_NONFINITE_INIT
    eng.set_state(np.ones((4, 1))); eng.run(5)            # log-density of x > 0 only: proposals at x <= 0 are rejected (NaN ratio)
    assert (eng.state()[0] > 0).all()
    eng.close()


def test_custom_target_through_the_job_api():
    """BasicContMuvParameter(:p, logtarget=..., gradlogtarget=...) with user closures through the host mirror: HMC on the
    curved 2-d density, every chain's saved history against the oracle."""
    tgt = K.CustomTarget(2, cases.SRC_BANANA)
    p = K.BasicContMuvParameter("p", logtarget=tgt)
    job = K.BasicMCJob(K.likelihood_model(p, False), K.HMC(0.15, 7), K.BasicMCRange(nsteps=300, burnin=50, thinning=2),
                       {"p": np.tile([0.5, 0.2], (96, 1))}, outopts={"diagnostics": ["accept"]}, seed=20260927)
    K.run(job)
    chain = K.output(job)
    assert chain.value(3).shape == (2, 125)
    assert job.range.nsteps == 299                                # last(51:2:300), BasicMCRange.jl:17
    o = O.OracleJob(sampler=L.SAMPLER_HMC, target_kind=L.TARGET_CUSTOM, nchains=96, ndims=2, nsteps=299, burnin=50, thinning=2,
                    leapstep=0.15, nleaps=7, custom_src=cases.SRC_BANANA, layout=job.engine.layout(), want_hist=True)
    o.set_state(np.tile([0.5, 0.2], (96, 1))); o.run(299)
    for c in (0, 3, 95):
        assert np.array_equal(chain.value(c), o.hist[:, c, :].T)
    assert np.array_equal(job.engine.accept_mask(), o.accept)
    # the sampler explores the curved ridge: E[x1 - x0^2] = 0 and Var[x1 - x0^2] = 1/2 under the target
    v = np.stack([chain.value(c) for c in range(96)])           # (chains, 2, n)
    r = v[:, 1, :] - v[:, 0, :] ** 2
    assert abs(r.mean()) < 0.1 and abs(r.var() - 0.5) < 0.1
    job.close()


@pytest.mark.parametrize("name,ring,pieces", [("mala_swiss", 7, [13, 20, 7]), ("dt_hmc_d100", 4, [20]), ("hmc_dense_d37", 5, [3, 12]), ("hmc_rats", 6, [30]),
                                              ("hmc_dense_d300_split", 3, [4, 6]), ("mala_dense_d512_split_mean_tuned", 4, [11, 19])])
def test_history_ring_keeps_the_last_saved_steps(name, ring, pieces):
    """klara_desc.hist_ring_cols: the history monitors keep a ring of the last R saved steps — after any sequence of runs the
    read-back equals the last R columns of the same job's full history, for value, logtarget and gradlogtarget."""
    case = cases.make_case(name)
    mon = L.MON_HISTORY | L.MON_HIST_LT | L.MON_HIST_GRAD
    full = K.Engine(**cases.engine_kwargs(case, monitor=mon)); part = K.Engine(**cases.engine_kwargs(case, monitor=mon, hist_ring_cols=ring, steps_per_launch=5))
    for e in (full, part):
        e.set_state(case["x0"]) if case["x0"] is not None else e.init_state_normal()
    done = 0
    for k in pieces:
        full.run(k); part.run(k); done += k
        nsaved = full.saved_steps()
        assert part.saved_steps() == nsaved
        keep = min(ring, nsaved)
        for c in (0, case["nchains"] - 1):
            v = part.chain(c)
            assert v.shape == (case["target"].ndims, keep)
            if keep:
                assert np.array_equal(v, full.chain(c)[:, nsaved - keep:])
                lt, g = part.chain_fields(c, True, True); flt, fg = full.chain_fields(c, True, True)
                assert np.array_equal(lt, flt[nsaved - keep:]) and np.array_equal(g, fg[:, nsaved - keep:])
    with pytest.raises(K.KlaraError):
        part.chain_mcvar(5, 0)                       # the post-hoc estimators need every saved step
    full.close(); part.close()


@pytest.mark.parametrize("name,maxlag,spl,mon", [("mala_d3_tuned", 9, 7, 0), ("hmc_d10_tuned_pooled", 6, 7, 0), ("dt_mala_d100_small_step", 15, 7, 0), ("mh_readme", 31, 7, 0),
                                                 # windows beyond 32 lags (lag blocks of 32; round 5): the estimator's own 32-column ring (launches of <= 32 saved samples: the delayed
                                                 # sequence comes from the kept tail), and with a value history and 50-transition launches (it comes from the launch's own columns too)
                                                 ("mh_readme", 100, 7, 0), ("mh_readme", 127, 50, L.MON_HISTORY), ("mala_d3_tuned", 40, 50, L.MON_HISTORY),
                                                 ("mala_dense_d512_split_mean_tuned", 5, 7, 0)])
def test_streaming_autocovariance_estimators(name, maxlag, spl, mon):
    """klara_desc.acov_maxlag: mcvar(:imse, maxlag) and mcvar(:ipse, maxlag) (mcvar.jl:75-105, 137-158) of every (chain, dimension)
    series from cross-products accumulated while sampling — no stored history — against (a) the NumPy restatement
    (klara_jl_amd.stats, FFT autocovariance) on individual chains of a full-history twin of the job, (b) the device's post-hoc
    estimators over that history.  Tolerance 1e-8 relative: three ways of summing the same products."""
    case = cases.make_case(name)
    n = {"mh_readme": 3000}.get(name, max(case["nsteps"], 6 * maxlag + case.get("burnin", 0)))
    case = dict(case, nsteps=n)
    twin = K.Engine(**cases.engine_kwargs(case, monitor=L.MON_HISTORY))
    eng = K.Engine(**cases.engine_kwargs(case, monitor=L.MON_SUMMARIES | mon, acov_maxlag=maxlag, steps_per_launch=spl))
    for e in (twin, eng):
        e.set_state(case["x0"]) if case["x0"] is not None else e.init_state_normal()
    twin.run(n)
    for k in (n // 3, n - n // 3):
        eng.run(k)
    imse, ipse, ns = eng.chain_acov_mcvar()
    assert ns == twin.saved_steps() > 2 * maxlag
    x1, _, _ = eng.state(); x2, _, _ = twin.state()
    assert np.array_equal(x1, x2)                                   # same job
    ph_imse = twin.chain_mcvar(5, maxlag, want=("imse",))[2]; ph_ipse = twin.chain_mcvar_ipse(maxlag)
    assert np.allclose(imse, ph_imse, rtol=1e-8, atol=1e-300) and np.allclose(ipse, ph_ipse, rtol=1e-8, atol=1e-300)
    for c in (0, case["nchains"] // 2, case["nchains"] - 1):
        v = twin.chain(c)
        for d in range(v.shape[0]):
            if v[d].std() == 0:
                continue
            assert np.isclose(imse[c, d], K.stats.mcvar(v[d], "imse", maxlag), rtol=1e-8), (c, d)
            assert np.isclose(ipse[c, d], K.stats.mcvar(v[d], "ipse", maxlag), rtol=1e-8), (c, d)
    twin.close(); eng.close()


def test_streamed_job_ess_default_uses_the_jobs_own_lag_window():
    """chain_ess / chain_iact / chain_mcvar(:imse | :ipse) with maxlag left at its default on a job that streams its autocovariances and
    stores no values use the job's own window (acov_maxlag, a documented truncation) instead of raising; an explicit other maxlag is refused."""
    p = K.BasicContMuvParameter("p", logtarget=K.GaussDiagTarget.negdot(3))
    job = K.BasicMCJob(K.likelihood_model(p, False), K.MH(np.full(3, 0.6)), K.BasicMCRange(nsteps=1100, burnin=100), {"p": np.zeros((70, 3))},
                       outopts={"destination": "none"}, seed=20260927, acov_maxlag=12)
    K.run(job)
    chain = K.output(job)
    ess, iact, imse = K.chain_ess(chain), K.chain_iact(chain, "ipse"), K.chain_mcvar(chain, "imse")
    assert np.array_equal(ess, K.chain_ess(chain, maxlag=12)) and np.array_equal(iact, K.chain_iact(chain, "ipse", maxlag=12))
    assert np.array_equal(imse, K.chain_mcvar(chain, "imse", maxlag=12)) and np.all(np.isfinite(ess)) and np.all(ess < 1000) and np.all(iact > 1.0)
    for bad in (0, 5, 31):
        with pytest.raises(ValueError, match="maxlag=12 only"):
            K.chain_ess(chain, maxlag=bad)
    job.close()


def test_iostream_sink_streams_with_bounded_memory(tmp_path):
    """:destination => :iostream with :flush (jobs.jl:17-29): the sink writes while the job runs — the device holds a ring of
    outopts chunk = 7 saved steps — and the files equal the ones written in one go from a full history."""
    from klara_jl_amd.iostream import write_chain
    case = cases.make_case("mala_swiss")
    X, y = cases.swiss_data()
    p = K.BasicContMuvParameter("p", logtarget=K.LogisticTarget(X, y, 100.0))
    job = K.BasicMCJob(K.likelihood_model(p, False), K.MALA(0.1), K.BasicMCRange(nsteps=40, burnin=10, thinning=2), {"p": case["x0"]},
                       outopts={"destination": "iostream", "filepath": str(tmp_path / "s"), "monitor": ["value", "logtarget", "gradlogtarget"],
                                "diagnostics": ["accept"], "flush": True, "chunk": 7}, seed=20260927)
    assert job.engine.chain(0).shape == (4, 0)
    K.run(job)
    assert job.engine.chain(0).shape == (4, 7)                       # the device never held more than the ring
    twin = K.Engine(sampler=L.SAMPLER_MALA, target=K.LogisticTarget(X, y, 100.0), nchains=70, nsteps=39, burnin=10, thinning=2, driftstep=0.1,
                    monitor=L.MON_HISTORY | L.MON_HIST_LT | L.MON_HIST_GRAD | L.MON_ACCEPT)
    twin.set_state(case["x0"]); twin.run(39)
    post = np.arange(11, 40, 2) - 1
    for c in (0, 41, 69):
        lt, g = twin.chain_fields(c, True, True)
        write_chain(str(tmp_path / "r" / f"chain_{c + 1:02d}"), "csv", twin.chain(c), lt, g, twin.accept_mask()[post, c])
        for f in ("value", "logtarget", "gradlogtarget", "diagnosticvalues"):
            a = (tmp_path / "s" / f"chain_{c + 1:02d}" / f"{f}.csv").read_text(); b = (tmp_path / "r" / f"chain_{c + 1:02d}" / f"{f}.csv").read_text()
            assert a == b and a.count("\n") == 15, (c, f)
    job.close(); twin.close()


def test_likelihood_prior_closures_through_the_job_api():
    """BasicContMuvParameter(:p, loglikelihood=..., logprior=..., gradloglikelihood=..., gradlogprior=...) with
    :monitor => [:value, :logtarget, :loglikelihood, :logprior] (BasicContMuvParameter.jl:174-201, iterate/MALA.jl:104-109): MALA on the
    Normal-Normal model of the reference's parameter tests.  Trajectory against the oracle (same closures compiled for the host);
    at every saved step logtarget == loglikelihood + logprior bit for bit, and the two parts equal the host closures at the saved values."""
    import ctypes as C
    case = cases.make_case("custom_normal_normal_mala")
    t = case["target"]
    p = K.BasicContMuvParameter("p", loglikelihood=cases.SRC_NN_LL, logprior=cases.SRC_NN_LP, gradloglikelihood=cases.SRC_NN_GLL,
                                gradlogprior=cases.SRC_NN_GLP, ndims=6, data=t.data)
    job = K.BasicMCJob(K.likelihood_model(p, False), K.MALA(0.4), K.BasicMCRange(nsteps=60, burnin=10), {"p": case["x0"]},
                       outopts={"monitor": ["value", "logtarget", "loglikelihood", "logprior"], "diagnostics": ["accept"]}, seed=20260927)
    K.run(job)
    chain = K.output(job)
    o = O.OracleJob(**cases.oracle_kwargs(case, layout=job.engine.layout()), want_hist=True)
    o.set_state(case["x0"]); o.run(60)
    assert np.array_equal(job.engine.accept_mask(), o.accept) and 0.2 < o.accept.mean() < 0.95
    lib = O.compile_user_target(t.source, 6)[0]
    dp = C.POINTER(C.c_double)
    for f in (lib.klara_user_loglikelihood, lib.klara_user_logprior):
        f.restype = C.c_double; f.argtypes = [dp, C.c_int, dp, C.c_longlong]
    for c in (0, 33, 69):
        v = chain.value(c); lt = chain.logtarget(c); ll = chain.loglikelihood(c); lp = chain.logprior(c)
        assert np.array_equal(v, o.hist[:, c, :].T) and np.array_equal(lt, o.hist_lt[:, c])
        assert np.array_equal(lt, ll + lp)
        for i in range(v.shape[1]):
            xi = np.ascontiguousarray(v[:, i])
            assert ll[i] == lib.klara_user_loglikelihood(xi.ctypes.data_as(dp), 6, t.data.ctypes.data_as(dp), t.data.size)
            assert lp[i] == lib.klara_user_logprior(xi.ctypes.data_as(dp), 6, t.data.ctypes.data_as(dp), t.data.size)
    # posterior mean of the conjugate model: (x / s + mu0 / s0) / (1 / s + 1 / s0)
    x, sv, m0, s0 = t.data[:6], t.data[6:12], t.data[12:18], t.data[18:24]
    post_mean = (x / sv + m0 / s0) / (1 / sv + 1 / s0); post_sd = np.sqrt(1 / (1 / sv + 1 / s0))
    m = K.mean(chain).mean(axis=0)
    assert np.all(np.abs(m - post_mean) < 0.35 * post_sd), (m, post_mean)
    with pytest.raises(ValueError):
        K.BasicMCJob(K.likelihood_model(K.BasicContMuvParameter("p", logtarget=K.GaussDiagTarget.negdot(3)), False), K.MALA(0.4),
                     K.BasicMCRange(nsteps=5), {"p": np.zeros((2, 3))}, outopts={"monitor": ["value", "loglikelihood"]})
    job.close()


def test_likelihood_prior_monitors_of_a_staged_closure():
    """The same monitors beyond 32 dimensions, where the closures are evaluated from the chain's rows of LDS by every lane of the chain
    (klara_custom.h STAGED, three rows for the likelihood + prior form): logtarget == loglikelihood + logprior bit for bit at every saved
    step, both parts equal the host closures at the saved values, accept masks and histories equal the oracle's."""
    import ctypes as C
    case = cases.make_case("staged_normal_normal_mala_d48")
    t = case["target"]
    d = t.ndims
    p = K.BasicContMuvParameter("p", loglikelihood=cases.SRC_NN_LL, logprior=cases.SRC_NN_LP, gradloglikelihood=cases.SRC_NN_GLL,
                                gradlogprior=cases.SRC_NN_GLP, ndims=d, data=t.data)
    job = K.BasicMCJob(K.likelihood_model(p, False), K.MALA(0.1), K.BasicMCRange(nsteps=40, burnin=10), {"p": case["x0"]},
                       outopts={"monitor": ["value", "logtarget", "loglikelihood", "logprior"], "diagnostics": ["accept"]}, seed=20260927)
    assert job.engine.layout()[1] > 1                                   # several lanes per chain: the staged form
    K.run(job)
    chain = K.output(job)
    o = O.OracleJob(**cases.oracle_kwargs(case, layout=job.engine.layout()), want_hist=True)
    o.set_state(case["x0"]); o.run(40)
    assert np.array_equal(job.engine.accept_mask(), o.accept) and 0.1 < o.accept.mean() < 0.99
    lib = O.compile_user_target(t.source, d)[0]
    dp = C.POINTER(C.c_double)
    for f in (lib.klara_user_loglikelihood, lib.klara_user_logprior):
        f.restype = C.c_double; f.argtypes = [dp, C.c_int, dp, C.c_longlong]
    for c in (0, 33, 69):
        v = chain.value(c); lt = chain.logtarget(c); ll = chain.loglikelihood(c); lp = chain.logprior(c)
        assert np.array_equal(v, o.hist[:, c, :].T) and np.array_equal(lt, o.hist_lt[:, c])
        assert np.array_equal(lt, ll + lp)
        for i in range(v.shape[1]):
            xi = np.ascontiguousarray(v[:, i])
            assert ll[i] == lib.klara_user_loglikelihood(xi.ctypes.data_as(dp), d, t.data.ctypes.data_as(dp), t.data.size)
            assert lp[i] == lib.klara_user_logprior(xi.ctypes.data_as(dp), d, t.data.ctypes.data_as(dp), t.data.size)
    job.close()


def test_iostream_sink_writes_the_likelihood_and_prior_files(tmp_path):
    """:destination => :iostream with :monitor => [:value, :loglikelihood, :logprior] (BasicContParamIOStream.jl:64-82: one file per
    monitored field): loglikelihood.csv / logprior.csv are written chunk by chunk next to value.csv, their lines are the in-memory
    job's values, and the accept diagnostics come back through the windowed read (klara_get_accept_rows) — ADVICE r2."""
    from klara_jl_amd.iostream import julia_float_repr as j
    case = cases.make_case("custom_normal_normal_mala")
    t = case["target"]
    mk = lambda: K.BasicContMuvParameter("p", loglikelihood=cases.SRC_NN_LL, logprior=cases.SRC_NN_LP, gradloglikelihood=cases.SRC_NN_GLL,
                                         gradlogprior=cases.SRC_NN_GLP, ndims=6, data=t.data)
    rng_ = K.BasicMCRange(nsteps=47, burnin=10, thinning=3)
    mem = K.BasicMCJob(K.likelihood_model(mk(), False), K.MALA(0.4), rng_, {"p": case["x0"]},
                       outopts={"monitor": ["value", "loglikelihood", "logprior"], "diagnostics": ["accept"]}, seed=20260927)
    K.run(mem); chain = K.output(mem)
    job = K.BasicMCJob(K.likelihood_model(mk(), False), K.MALA(0.4), rng_, {"p": case["x0"]},
                       outopts={"destination": "iostream", "filepath": str(tmp_path / "s"), "monitor": ["value", "loglikelihood", "logprior"],
                                "diagnostics": ["accept"], "chunk": 5}, seed=20260927)
    K.run(job)
    post = np.arange(11, 48, 3) - 1
    rows = mem.engine.accept_rows(int(post[2]), int(post[6] - post[2] + 1))
    assert np.array_equal(rows, mem.engine.accept_mask()[post[2]:post[6] + 1])
    for c in (0, 41, 69):
        d = tmp_path / "s" / f"chain_{c + 1:02d}"
        assert (d / "loglikelihood.csv").read_text().splitlines() == [j(v) for v in chain.loglikelihood(c)]
        assert (d / "logprior.csv").read_text().splitlines() == [j(v) for v in chain.logprior(c)]
        assert (d / "value.csv").read_text().splitlines() == [",".join(j(v) for v in chain.value(c)[:, i]) for i in range(len(post))]
        assert (d / "diagnosticvalues.csv").read_text().splitlines() == ["true" if a else "false" for a in mem.engine.accept_mask()[post, c]]
    job.close(); mem.close()


# ------------------------------------------------------------------ full-size parity on sampled chains
@pytest.mark.parametrize("name,kw,nsteps", [
    ("cfg2_mala", dict(sampler=L.SAMPLER_MALA, driftstep=0.9), 40),
    ("cfg3_hmc_dense", dict(sampler=L.SAMPLER_HMC, leapstep=0.1, nleaps=10), 6),
    ("hmc_iso", dict(sampler=L.SAMPLER_HMC, leapstep=0.1, nleaps=10), 12),
    ("staged_closure_mala", dict(sampler=L.SAMPLER_MALA, driftstep=0.3), 12),       # the README closure as a whole-vector user closure: 8 lanes per chain, staged through LDS
    ("pair_closure_hmc", dict(sampler=L.SAMPLER_HMC, leapstep=0.1, nleaps=5), 8),   # ... and as a pair closure on the pair-transposed layout
])
def test_full_size_bit_exact_on_sampled_chains(name, kw, nsteps):
    """BASELINE.json sizes (65,536 chains x 100 dims): chains are independent and the stream is keyed by the global
    chain id, so the oracle can replay ANY block of chains of the full-size GPU job.  Three blocks of 16 chains
    (first, middle, last — the last one exercises the partially filled final wavefront) are compared bit for bit."""
    n, d = 65536 - 5, 100                               # not a multiple of the chains-per-wavefront: ragged tail
    target = K.GaussDenseTarget.compound_symmetric(d, 0.5) if name == "cfg3_hmc_dense" else K.GaussDiagTarget.negdot(d)
    if name == "staged_closure_mala":
        target = K.CustomTarget(d, cases.SRC_NEGDOT)
    elif name == "pair_closure_hmc":
        target = K.CustomTarget.pairwise(d, cases.SRC_PAIR_NEGDOT)
    eng = K.Engine(target=target, nchains=n, nsteps=nsteps, monitor=L.MON_ACCEPT | L.MON_SUMMARIES, steps_per_launch=3, **kw)
    if name == "staged_closure_mala":
        assert tuple(eng.layout()) == (0, 8, 14)
    eng.init_state_normal()
    eng.run(nsteps)
    x, lt, g = eng.state()
    mask = eng.accept_mask()
    s, q, _ = eng.chain_sums()
    case = dict(kw, target=target, nchains=16, nsteps=nsteps, name=name, x0=None, seed=20260927)
    for off in (0, 32768 + 7, n - 16):
        job = O.OracleJob(**cases.oracle_kwargs(case, layout=eng.layout(), chain_offset=off))
        job.init_state_normal(); job.run(nsteps)
        sl = slice(off, off + 16)
        assert np.array_equal(mask[:, sl], job.accept), (name, off)
        assert np.array_equal(x[sl], job.X) and np.array_equal(lt[sl], job.LT) and np.array_equal(g[sl], job.G), (name, off)
        assert np.array_equal(s[sl], job.sum) and np.array_equal(q[sl], job.sumsq), (name, off)
    eng.close()


@pytest.mark.parametrize("name,kw,nsteps,spl", [
    ("cfg2_mala_1step", dict(sampler=L.SAMPLER_MALA, driftstep=0.9), 30, 1),
    ("mala_small_step_fused", dict(sampler=L.SAMPLER_MALA, driftstep=0.05), 24, 8),
    ("hmc_iso_1step", dict(sampler=L.SAMPLER_HMC, leapstep=0.1, nleaps=10), 8, 1),
])
def test_full_size_pair_transposed_on_sampled_chains(name, kw, nsteps, spl):
    """The bench configuration itself (65,536-chain shape, nothing monitored but the accept mask -> layout kind 3, two
    chain partitions on two streams): blocks of 16 chains at the start, across the partition boundary and in the ragged
    last wavefront group are replayed by the oracle and compared bit for bit."""
    n, d = 65536 - 5, 100
    target = K.GaussDiagTarget.negdot(d)
    eng = K.Engine(target=target, nchains=n, nsteps=nsteps, monitor=L.MON_ACCEPT, steps_per_launch=spl, **kw)
    assert eng.layout()[0] == 3
    eng.init_state_normal()
    eng.run(nsteps)
    x, lt, g = eng.state()
    mask = eng.accept_mask()
    na, _ = eng.accept_counts()
    assert np.array_equal(na, mask.sum(axis=0))
    blocks = (n + 15) // 16                               # partitions are cut in blocks of 16 chains (klara_api.hip part_range)
    boundary = ((blocks + 1) // 2) * 16                   # first chain of the second partition
    case = dict(kw, target=target, nchains=16, nsteps=nsteps, name=name, x0=None, seed=20260927)
    for off in (0, boundary - 8, n - 16):
        job = O.OracleJob(**cases.oracle_kwargs(case, layout=eng.layout(), chain_offset=off))
        job.init_state_normal(); job.run(nsteps)
        sl = slice(off, off + 16)
        assert np.array_equal(mask[:, sl], job.accept), (name, off)
        assert np.array_equal(x[sl], job.X) and np.array_equal(lt[sl], job.LT) and np.array_equal(g[sl], job.G), (name, off)
    eng.close()


@pytest.mark.parametrize("sparse", [0, 1, 2])
def test_full_size_bench_job_replayed_by_the_oracle(sparse):
    """The job bench.py times, as the driver runs it: MALA driftstep 0.9 on lt = -|x|^2, D = 100, 65,531 chains (ragged last group),
    running sums on, the library's 32 transitions per launch, seed 20260927; `run(5)` (the driver's warm-up) and `run(20)` (its timed
    region: ONE launch, whole job on the caller's stream), then 96 more transitions (three launches on two chain partitions / streams).
    sparse = 0 is what bench.py runs — the kernel of every launch chosen on the device from the previous launch's accepted proposals:
    resident sums (8 lanes) while the fresh job still accepts tens of per cent, atomic folds (4 lanes) once it has settled —, 1 and 2
    pin the two kernel families.  Blocks of 16 chains at the start, across the partition boundary and in the ragged tail are replayed
    by the oracle: accept masks of all 121 transitions, x / logtarget / gradient, running sums — bit for bit in all three modes."""
    n, d, runs = 65536 - 5, 100, (5, 20, 96)
    nsteps = sum(runs)
    target = K.GaussDiagTarget.negdot(d)
    kw = dict(sampler=L.SAMPLER_MALA, driftstep=0.9)
    eng = K.Engine(target=target, nchains=n, nsteps=nsteps, burnin=0, seed=20260927, monitor=L.MON_ACCEPT | L.MON_SUMMARIES,
                   sparse_moves=sparse, **kw)
    assert eng.layout() == (3, 8, 14)
    eng.init_state_normal()
    for r in runs:
        eng.run(r)
    cnt, last_mode, last_acc = eng.launch_modes()
    assert int(cnt.sum()) == 5                                      # 1 + 1 + 3 launches
    if sparse == 0:
        assert cnt[2] == 5, cnt                                     # every launch a device-decided pair (5, 20, 32, 32, 32 transitions)
        assert last_mode[0] in (0, 1) and last_acc[0] >= 0
    else:
        assert cnt[sparse - 1] == 5, cnt
    x, lt, g = eng.state()
    mask = eng.accept_mask()
    s, q, nsaved = eng.chain_sums()
    assert nsaved == nsteps
    na, _ = eng.accept_counts()
    assert np.array_equal(na, mask.sum(axis=0))
    if sparse == 0:
        # the fresh job (x0 ~ N(0, I)) accepts tens of per cent of its first proposals and about 1 % after a hundred transitions:
        # the device-side decision must have left the 8-lane kernels behind by the last launch
        assert mask[:5].mean() > 0.2 and mask[-32:].mean() < 0.04, (mask[:5].mean(), mask[-32:].mean())
        assert last_mode[0] == 0 and last_mode[1] == 0, last_mode
    blocks = (n + 15) // 16
    boundary = ((blocks + 1) // 2) * 16
    case = dict(kw, target=target, nchains=16, nsteps=nsteps, burnin=0, name="bench_job", x0=None, seed=20260927)
    for off in (0, boundary - 8, n - 16):
        job = O.OracleJob(**cases.oracle_kwargs(case, layout=eng.layout(), chain_offset=off))
        job.init_state_normal(); job.run(nsteps)
        sl = slice(off, off + 16)
        assert np.array_equal(mask[:, sl], job.accept), off
        assert np.array_equal(x[sl], job.X) and np.array_equal(lt[sl], job.LT) and np.array_equal(g[sl], job.G), off
        assert np.array_equal(s[sl], job.sum) and np.array_equal(q[sl], job.sumsq), off
    eng.close()


def test_kernel_attributes_and_launch_modes_queries():
    """klara_get_kernel_attributes reports registers / scratch / static LDS of the kernel a handle launches, from the loaded code
    object and without launching anything (bench.py checks its committed PMC summaries against it); klara_get_launch_modes counts how the
    launches were issued."""
    neg = K.GaussDiagTarget.negdot(100)
    e = K.Engine(sampler=L.SAMPLER_MALA, target=neg, nchains=4099, nsteps=10 ** 6, driftstep=0.9, monitor=L.MON_SUMMARIES)
    e.init_state_normal()
    before = e.state()[0].copy()
    four, eight, one = e.kernel_attributes(0, 32), e.kernel_attributes(1, 32), e.kernel_attributes(0, 1)
    assert np.array_equal(e.state()[0], before) and tuple(e.launch_modes()[0]) == (0, 0, 0)        # nothing ran, nothing was counted
    assert 128 < four[0] <= 168 and 168 < eight[0] <= 256 and four[2] >= 8192 and eight[2] >= 8192   # 3 / 2 wavefronts per SIMD; math tables in LDS
    assert one[0] > 0
    assert e.shader_clock_mhz() == 0.0                                                             # nothing launched yet
    e.run(64)
    assert tuple(e.launch_modes()[0]) == (0, 0, 2)                                                 # two device-decided launches
    assert 1200.0 < e.shader_clock_mhz() < 2700.0                                                  # the in-kernel clock probe of the last launch
    e.close()
    for kw, lo, hi in ((dict(sampler=L.SAMPLER_HMC, target=K.GaussDenseTarget.compound_symmetric(100, 0.5), leapstep=0.1, nleaps=10), 200, 256),
                       (dict(sampler=L.SAMPLER_MH, target=K.GaussDiagTarget.negdot(2), mh_sigma=[1.0, 1.0]), 16, 128),
                       (dict(sampler=L.SAMPLER_MALA, target=K.CustomTarget.pairwise(100, cases.SRC_PAIR_NEGDOT), driftstep=0.5), 64, 256)):
        e = K.Engine(nchains=300, nsteps=100, **kw)
        v, sc, lds = e.kernel_attributes(0, 32)
        assert lo <= v <= hi and sc >= 0 and lds >= 0, (kw["sampler"], v, sc, lds)
        assert e.kernel_attributes(1, 32) == (v, sc, lds)                                          # one kernel family: the same kernel
        assert tuple(e.launch_modes()[0]) == (0, 0, 0)
        e.close()


def test_pair_transposed_slice_sampler_moments():
    """Slice sampler on layout kind 3, MvNormal(mu, sigma) with D = 20, 16,384 chains x 60 transitions from x0 ~ N(0, I): the
    ensemble of final states has the target's mean and standard deviation (tolerance 5 standard errors, max over coordinates)."""
    n, d = 16384, 20
    mu = np.linspace(-1, 2, d); sg = np.linspace(0.5, 2.0, d)
    eng = K.Engine(sampler=L.SAMPLER_SLICE, target=K.GaussDiagTarget.mvnormal(mu, sg), nchains=n, nsteps=60, slice_widths=2.0 * sg,
                   monitor=0, steps_per_launch=10)
    assert eng.layout()[0] == 3
    eng.init_state_normal(); eng.run(60)
    x, lt, _ = eng.state()
    assert np.allclose(lt, -0.5 * (((x - mu) / sg) ** 2).sum(axis=1) - 0.5 * d * np.log(2 * np.pi) - np.log(sg).sum(), rtol=1e-12)
    assert np.max(np.abs(x.mean(axis=0) - mu) / sg) < 5 / np.sqrt(n), np.max(np.abs(x.mean(axis=0) - mu) / sg)
    assert np.max(np.abs(x.std(axis=0) / sg - 1.0)) < 5 / np.sqrt(2 * n), np.max(np.abs(x.std(axis=0) / sg - 1.0))
    eng.close()


def test_full_size_pair_transposed_hmc_moments():
    """65,536 chains x 100 dims on layout kind 3, HMC L=10 eps=0.1 (mixes in a few transitions): the ensemble of final
    states has the target's moments (mean 0, var 1/2 per coordinate; tolerances = 5 standard errors of 65,536 draws,
    maximum over 100 coordinates), lt and the gradient are consistent with x, and one stream or two give the same bits."""
    n, d = 65536, 100
    kw = dict(sampler=L.SAMPLER_HMC, target=K.GaussDiagTarget.negdot(d), nchains=n, nsteps=60, leapstep=0.1, nleaps=10, monitor=0)
    a = K.Engine(steps_per_launch=16, **kw); a.init_state_normal(); a.run(60)
    b = K.Engine(steps_per_launch=1, nstreams=1, **kw); b.init_state_normal(); b.run(60)
    assert a.layout()[0] == 3
    xa, lta, ga = a.state(); xb, ltb, gb = b.state()
    assert np.array_equal(xa, xb) and np.array_equal(lta, ltb) and np.array_equal(ga, gb)
    assert np.allclose(lta, -(xa * xa).sum(axis=1), rtol=1e-12) and np.array_equal(ga, -2.0 * xa)
    mean = xa.mean(axis=0); var = xa.var(axis=0)
    assert np.max(np.abs(mean)) < 5 * np.sqrt(0.5 / n), np.max(np.abs(mean))
    assert np.max(np.abs(var - 0.5)) < 5 * 0.5 * np.sqrt(2.0 / n), (var.min(), var.max())
    na, nt = a.accept_counts()
    assert nt == 60 and 0.8 < na.mean() / 60 <= 1.0
    a.close(); b.close()


# ------------------------------------------------------------------ full-size properties (BASELINE shapes)
def test_full_size_mala_properties():
    """BASELINE cfg 2 shape (65,536 chains x 100 dims): determinism, launch-split invariance, and pooled
    posterior moments.  Truth for lt = -|x|^2: mean 0, var 1/2.  The drift step is tuned per chain
    (AcceptanceRate 0.574) because drift 0.9 accepts ~2% at D=100 and has not mixed in a test-sized run;
    tolerance 1e-2 on var (Monte-Carlo error of 65,536 x 200 correlated samples), 5e-3 on mean."""
    n, d = 65536, 100
    kw = dict(sampler=L.SAMPLER_MALA, target=K.GaussDiagTarget.negdot(d), nchains=n, nsteps=500, burnin=300,
              driftstep=0.9, tuner=L.TUNER_ACCEPT_RATE, targetrate=0.574, period=25, monitor=L.MON_SUMMARIES)
    a = K.Engine(steps_per_launch=25, **kw); a.init_state_normal(); a.run(500)
    b = K.Engine(steps_per_launch=4, **kw); b.init_state_normal(); b.run(123); b.run(377)
    xa, lta, _ = a.state(); xb, ltb, _ = b.state()
    assert np.array_equal(xa, xb) and np.array_equal(lta, ltb)
    s, q, na, nt, ns = a.pooled_summaries()
    cnt = ns * n
    mean = s / cnt; var = q / cnt - mean * mean
    assert np.max(np.abs(mean)) < 5e-3, np.max(np.abs(mean))
    assert np.max(np.abs(var - 0.5)) < 1e-2, (var.min(), var.max())
    step = a.tune()[0]
    assert 0.05 < np.median(step) < 0.9 and 0.4 < na / nt < 0.75, (np.median(step), na / nt)
    a.close(); b.close()


def test_full_size_hmc_dense_properties():
    """BASELINE cfg 3 shape (65,536 chains, D=100, dense precision, L=10, eps=0.1): the FP64-MFMA path.
    Energy error of a leapfrog trajectory is O(eps^2): acceptance > 0.9; starting from x0 ~ N(0, I) the
    pooled variance moves towards the truth (1.0 on the diagonal); lt stays consistent with x:
    lt == -1/2 x'Px recomputed on the host to 1e-10 relative."""
    n, d = 65536, 100
    t = K.GaussDenseTarget.compound_symmetric(d, 0.5)
    eng = K.Engine(sampler=L.SAMPLER_HMC, target=t, nchains=n, nsteps=40, burnin=20, leapstep=0.1, nleaps=10,
                   monitor=L.MON_SUMMARIES, steps_per_launch=5)
    eng.init_state_normal(); eng.run(40)
    x, lt, g = eng.state()
    ref_lt = -0.5 * np.einsum("ni,ij,nj->n", x[:512], t.precision, x[:512])
    assert np.allclose(lt[:512], ref_lt, rtol=1e-10, atol=1e-10)
    assert np.allclose(g[:512], -x[:512] @ t.precision, rtol=1e-10, atol=1e-10)
    s, q, na, nt, ns = eng.pooled_summaries()
    assert na / nt > 0.9
    mean = s / (ns * n)
    assert np.max(np.abs(mean)) < 0.02
    eng.close()


def test_c_example_runs_the_readme_job(tmp_path):
    """examples/readme_job.c (plain C99 against the C ABI): the README job for 4,096 chains reproduces the target's moments."""
    import re, subprocess
    from test_host_api import _build_c_example
    r = subprocess.run([str(_build_c_example(tmp_path)), "4096"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    m = re.search(r"mean = \(([-0-9.]+), ([-0-9.]+)\).*E\[z1\^2\] = ([0-9.]+).*acceptance = ([0-9.]+)", r.stdout)
    assert m, r.stdout
    m0, m1, v0, acc = map(float, m.groups())
    assert abs(m0) < 0.01 and abs(m1) < 0.01 and abs(v0 - 0.5) < 0.01 and 0.35 < acc < 0.5
    p = re.search(r"pooled over (\d+) samples: mean = \(([-0-9.]+), ([-0-9.]+)\), var = \(([0-9.]+), ([0-9.]+)\).*acceptance = ([0-9.]+)", r.stdout)
    assert p, r.stdout                                         # klara_gather_moments from plain C
    ns, pm0, pm1, pv0, pv1, pacc = int(p.group(1)), *map(float, p.groups()[1:])
    assert ns == 4096 * 9000 and abs(pm0 - m0) < 1e-4 and abs(pm1 - m1) < 1e-4 and abs(pv0 - 0.5) < 0.01 and abs(pv1 - 0.5) < 0.01 and abs(pacc - acc) < 1e-3
