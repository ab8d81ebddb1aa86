"""CPU tests: pin the oracle (and the shared deterministic primitives) against every known-answer the
reference's own tests hold for this path and against published vectors."""
import math

import os

import numpy as np
import pytest
from scipy import stats

import cases
import oracle_ffi as O
from klara_jl_amd import _lib as L


def test_philox4x32_10_random123_kat():
    # Random123 kat_vectors (philox4x32 10 rounds): the published known-answer vectors
    assert O.philox_block([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert O.philox_block([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert O.philox_block([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_stream_block_counter_layout():
    # kd_stream_block = Philox with key=seed, counter=(t<<24|slot, chain): rocRAND's (seed, subsequence, offset/4)
    seed, chain, t, slot = 0x0123456789abcdef, 0x1_0000_0005, 7, 51
    blk = (t << 24) | slot
    ref = O.philox_block([blk & 0xffffffff, blk >> 32, chain & 0xffffffff, chain >> 32],
                         [seed & 0xffffffff, seed >> 32])
    assert list(O.stream_blocks(seed, chain, t, [slot])[0]) == ref


def test_uniform_is_exact_and_open(oracle):
    assert oracle.ko_u52(0, 0) == 2.0 ** -53
    assert oracle.ko_u52(0xffffffff, 0xffffffff) == 1.0 - 2.0 ** -53
    assert oracle.ko_u52(0x80000000, 0) == 0.5 + 2.0 ** -53


def _ulps(a, b):
    return np.max(np.abs(a - b) / np.spacing(np.abs(b)))


def test_detmath_accuracy_vs_libm():
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.random(200000), np.exp(rng.uniform(-700, 700, 200000)), 1 + rng.uniform(-1e-3, 1e-3, 20000)])
    assert _ulps(O.math_op(0, x), np.log(x)) <= 1.0
    x = np.concatenate([rng.uniform(-700, 700, 200000), rng.uniform(-1, 1, 200000)])
    assert _ulps(O.math_op(1, x), np.exp(x)) <= 1.0
    # the uniforms' own log (table method, detmath.h kd_log_u01): every kd_u52 value is a positive normal number
    m = np.concatenate([rng.integers(0, 2 ** 52, 400000), [0, 1, 2 ** 52 - 1, 2 ** 51, 2 ** 51 - 1, 2 ** 52 - 2 ** 30]]).astype(np.uint64)
    u = (m.astype(np.float64) + 0.5) * 2.0 ** -52          # exactly the kd_u52 lattice
    assert _ulps(O.math_op(7, u), np.log(u)) <= 1.0
    x = np.concatenate([np.exp(rng.uniform(-700, 700, 100000)), 1 + rng.uniform(-2e-2, 2e-2, 100000)])
    assert _ulps(O.math_op(7, x), np.log(x)) <= 1.0
    # sin/cos(2 pi u) on the lattice (table + rotation): absolute error < 2^-52
    ul = u.astype(np.longdouble)
    pi_l = np.longdouble("3.14159265358979323846264338327950288")
    assert np.max(np.abs(O.math_op(2, u) - np.sin(2 * pi_l * ul).astype(np.float64))) <= 2.3e-16
    assert np.max(np.abs(O.math_op(3, u) - np.cos(2 * pi_l * ul).astype(np.float64))) <= 2.3e-16
    s2, c2 = O.math_op(2, u), O.math_op(3, u)
    assert np.max(np.abs(s2 * s2 + c2 * c2 - 1.0)) <= 5e-16
    sp = O.math_op(0, [0.0, -1.0, np.inf, 1.0])
    assert sp[0] == -np.inf and np.isnan(sp[1]) and sp[2] == np.inf and sp[3] == 0.0
    se = O.math_op(1, [0.0, -1e9, 1e9, -745.0])
    assert se[0] == 1.0 and se[1] == 0.0 and se[2] == np.inf and se[3] == 5e-324


def test_softplus_logistic_pair():
    """The functions of the logistic targets' data rows (doc/examples/swiss/MALA/analytical.jl:13,17) against 160-bit arithmetic:
    kd_exp_neg(a) = exp(-a) within 0.51 ulp up to a = 708 (Arm-style reduction: round-to-nearest k from the low mantissa bits of one fma),
    clamped to exp(-708) beyond; kd_log12 on [1, 2]: absolute error below 1.5 * 2^-53, log12(1) = 0 exactly, full relative accuracy towards 1;
    log(1 + exp(x)) and 1 / (1 + exp(-x)) from ONE exponential: 2 ulp of max(value, 1) for the first, 2 ulp for the second; never negative,
    exact limits for large |x|, NaN passed through."""
    import mpmath
    mpmath.mp.prec = 160
    rng = np.random.default_rng(7)
    a = np.concatenate([rng.uniform(0, 708, 200000), rng.uniform(0, 40, 100000), rng.uniform(0, 1, 50000), [0.0, 708.0, 1e-300, 37.5]])
    got = O.math_op(9, a)
    assert np.all(np.abs(got - np.exp(-a)) <= np.spacing(np.exp(-a)))                  # within 1 ulp of libm everywhere
    worst = 0.0
    for ai, gi in zip(a[::53], got[::53]):
        e = mpmath.exp(-mpmath.mpf(float(ai)))
        worst = max(worst, float(abs(mpmath.mpf(float(gi)) - e) / mpmath.mpf(float(np.spacing(float(e))))))
    assert worst <= 0.51, worst
    assert O.math_op(9, np.array([0.0]))[0] == 1.0
    assert np.all(O.math_op(9, np.array([708.0000001, 745.0, 1e9, np.inf])) == O.math_op(9, np.array([708.0]))[0])     # clamped, a normal number
    assert 3.3e-308 < O.math_op(9, np.array([708.0]))[0] < 3.4e-308
    # kd_log12
    x12 = np.concatenate([1.0 + rng.random(200000), 1.0 + rng.random(50000) * 2.0 ** -7, 1.0 + 2.0 ** -rng.uniform(8, 52, 20000),
                          [1.0, 2.0, 1.0 + 2.0 ** -52, 2.0 - 2.0 ** -52, 1.0078125, 1.0078125 - 2.0 ** -52, 1.375]])
    l12 = O.math_op(12, x12)
    assert l12[np.flatnonzero(x12 == 1.0)[0]] == 0.0 and np.all(l12 >= 0.0)
    worst_abs = worst_rel_near_one = 0.0
    for xi, li in zip(x12[::41].tolist() + x12[-7:].tolist(), l12[::41].tolist() + l12[-7:].tolist()):
        t = mpmath.log(mpmath.mpf(xi))
        worst_abs = max(worst_abs, float(abs(mpmath.mpf(li) - t) * mpmath.mpf(2) ** 53))
        if 1.0 < xi < 1.0078125:
            worst_rel_near_one = max(worst_rel_near_one, float(abs(mpmath.mpf(li) - t) / t * mpmath.mpf(2) ** 53))
    assert worst_abs <= 1.5, worst_abs
    assert worst_rel_near_one <= 3.0, worst_rel_near_one       # (bin 0: r = x - 1 exactly; the dropped r^8/8 is 2^-52 of the value at its far end)
    x = np.concatenate([rng.uniform(-40, 40, 4000), rng.uniform(-750, 750, 500), [0.0, -0.0, 1e-320, 800.0, -800.0, 36.0, -36.5, 37.0, 1e-9, -1e-9]])
    sp, lg = O.math_op(10, x), O.math_op(11, x)
    assert np.all(sp >= 0.0) and np.all(lg >= 0.0) and np.all(lg <= 1.0)
    for xi, s_, l_ in zip(x, sp, lg):
        e = mpmath.exp(mpmath.mpf(float(xi)))
        ts, tl = mpmath.log(1 + e), e / (1 + e)
        # (log(1 + t) is formed from the rounded 1 + t, as the reference's log(1 + exp(Xp)) is: absolute error up to an ulp of 1)
        assert abs(mpmath.mpf(float(s_)) - ts) <= 2 * mpmath.mpf(float(np.spacing(max(abs(float(s_)), 1.0)))), xi
        if abs(xi) <= 708:
            assert abs(mpmath.mpf(float(l_)) - tl) <= 2 * mpmath.mpf(float(np.spacing(abs(float(l_)) or 5e-324))), xi
        else:
            assert l_ == (1.0 if xi > 0 else O.math_op(9, np.array([708.0]))[0])
    assert np.isnan(O.math_op(10, [np.nan])[0]) and np.isnan(O.math_op(11, [np.nan])[0])


def test_box_muller_moments():
    n = 200000
    blocks = O.stream_blocks(12345, 3, 0, range(n // 2))        # both halves of every block: n word pairs
    out = np.zeros((n, 2))
    words = np.ascontiguousarray(blocks.reshape(n, 2))
    O.load().ko_normal_pairs_w(n, words.ctypes.data, out.ctypes.data)
    z = out.ravel()
    assert abs(z.mean()) < 4 / math.sqrt(z.size)
    assert abs(z.var() - 1) < 0.01
    assert abs(np.corrcoef(out[:, 0], out[:, 1])[0, 1]) < 0.01
    assert abs(np.corrcoef(out[0::2, 0], out[1::2, 0])[0, 1]) < 0.01          # the two pairs of one block
    assert stats.kstest(z[:50000], "norm").pvalue > 1e-3


def _words_from_mantissas(m44, j20):
    """Word pairs (wa, wb) whose radius uniform has the 44-bit mantissa m44 and whose angle is direction j20 of 2^20 (bits 19..12 of wb,
    which neither uses, are left zero / set: they must not matter)."""
    m44 = np.asarray(m44, dtype=np.uint64); j20 = np.asarray(j20, dtype=np.uint64)
    w = np.empty((m44.size, 2), np.uint32)
    w[:, 0] = (m44 >> np.uint64(12)).astype(np.uint32)
    w[:, 1] = ((j20 << np.uint64(12)) | (m44 & np.uint64(0xfff))).astype(np.uint32)
    return w


def test_normal_pair_against_independent_high_precision_box_muller():
    """kd_normal_pair_w (detmath.h: the one source of normals on the device AND in the oracle) against an independent evaluation of the
    same definition from the same words: u1 = (2 m + 1) 2^-45 (m: 44 bits), u2 = (j + 1/2) 2^-20 + 2^-53 (j: 20 bits; cell centres),
    z0 = sqrt(-2 ln u1) cos(2 pi u2), z1 = sqrt(-2 ln u1) sin(2 pi u2) in mpmath at 160 bits — 100,000 random word pairs plus the
    extreme points of the lattice (smallest / largest radius uniform, directions at and next to every multiple of pi/2).

    Bound: |z - exact| <= 3 ulp(radius) (worst of the sample: 2.54).  ulp(z) is the wrong yardstick next to the zeros of sin / cos: kd_sincos2pi is accurate to
    2^-52 ABSOLUTE, so where |cos| ~ 1e-15 the product is off by a few 1e-16 — 30 % of a value that is itself 1e-15 of a standard
    deviation.  Where the trigonometric factor is >= 1/2 in magnitude the error is also asserted in ulps of z itself (<= 4.5: two
    binades of radius ulps)."""
    import mpmath as mp
    rng = np.random.default_rng(20260927)
    n = 100000
    top, jtop = (1 << 44) - 1, (1 << 20) - 1
    edge1 = [0, 1, 2, top, top - 1, 1 << 43, (1 << 43) - 1, 1 << 32, 12345]
    edge2 = [0, 1, jtop, jtop - 1] + [q * (1 << 18) + d for q in (1, 2, 3) for d in (-2, -1, 0, 1)] + [(j << 12) + d for j in (1, 77, 128, 255) for d in (-1, 0)]
    m1 = np.concatenate([rng.integers(0, 1 << 44, n, dtype=np.uint64), np.repeat(np.array(edge1, np.uint64), len(edge2))])
    j2 = np.concatenate([rng.integers(0, 1 << 20, n, dtype=np.uint64), np.tile(np.array(edge2, np.uint64), len(edge1))])
    w = _words_from_mantissas(m1, j2)
    out = np.zeros((w.shape[0], 2))
    O.load().ko_normal_pairs_w(w.shape[0], w.ctypes.data, out.ctypes.data)
    worst_rad = worst_z = 0.0
    with mp.workprec(160):
        two45, two20, two53, twopi = mp.mpf(2) ** -45, mp.mpf(2) ** -20, mp.mpf(2) ** -53, 2 * mp.pi
        for i in range(w.shape[0]):
            u1 = (2 * int(m1[i]) + 1) * two45; u2 = (int(j2[i]) + mp.mpf(1) / 2) * two20 + two53
            rad = mp.sqrt(-2 * mp.log(u1)); a = twopi * u2
            ulp_rad = mp.mpf(float(np.spacing(float(rad))))
            for h, trig in ((0, mp.cos(a)), (1, mp.sin(a))):
                err = abs(mp.mpf(float(out[i, h])) - rad * trig)
                worst_rad = max(worst_rad, float(err / ulp_rad))
                if abs(trig) >= 0.5:
                    worst_z = max(worst_z, float(err / mp.mpf(float(np.spacing(abs(float(rad * trig)))))))
    assert worst_rad <= 3.0, worst_rad
    assert worst_z <= 4.5, worst_z
    # the largest normal the generator can produce: u1 = 2^-45 -> sqrt(2 * 45 ln 2) = 7.898... along the direction nearest an axis (half a cell off it)
    assert abs(np.abs(out[n:]).max() - math.sqrt(90 * math.log(2)) * math.cos(2 * math.pi * (2.0 ** -21 + 2.0 ** -53))) < 1e-13
    # the accept uniform is the radius uniform of words (x, y): never 0, never 1, log above the kernels' skip guard (-31.2)
    assert O.load().ko_u44(0, 0) == 2.0 ** -45 and O.load().ko_u44(0xFFFFFFFF, 0xFFFFFFFF) == 1.0 - 2.0 ** -45
    assert O.math_op(7, np.array([2.0 ** -45]))[0] > -31.2


def test_draw_schedule_of_a_transition():
    """Which bits make which normal (detmath.h kd_normal_pair_at): element pair p takes half (p >> 3) & 1 of block slot
    (p & 7) + 8 (p >> 4) of its transition — restated here in Python and compared BIT FOR BIT with what the oracle's samplers draw
    (ko_transition_normals), for vector lengths on both sides of every boundary of the mapping.  Every (block, half) is used by one pair only, every
    block slot is below ceil(D/2), and the accept uniform is the 44-bit uniform of words (x, y) of block slot ceil(D/2), which no pair touches."""
    import ctypes as C
    lib = O.load()
    seed, chain, t = 20260927, (1 << 33) + 5, 123456
    for D in (1, 2, 3, 15, 16, 17, 18, 31, 32, 33, 34, 100, 255, 256):
        P = (D + 1) // 2
        z = np.zeros(D); u = np.zeros(1)
        lib.ko_transition_normals(C.c_uint64(seed), C.c_uint64(chain), C.c_uint64(t), D, z.ctypes.data, u.ctypes.data)
        slots = [((p & 7) + 8 * (p >> 4), (p >> 3) & 1) for p in range(P)]
        assert len(set(slots)) == P and max(b for b, _ in slots) < P
        blocks = O.stream_blocks(seed, chain, t, sorted({b for b, _ in slots} | {P}))
        byslot = dict(zip(sorted({b for b, _ in slots} | {P}), blocks))
        words = np.array([[byslot[b][2 * h], byslot[b][2 * h + 1]] for b, h in slots], dtype=np.uint32)
        ref = np.zeros((P, 2))
        lib.ko_normal_pairs_w(P, words.ctypes.data, ref.ctypes.data)
        assert np.array_equal(z, ref.ravel()[:D]), D
        assert u[0] == lib.ko_u44(int(byslot[P][0]), int(byslot[P][1]))


def test_draw_schedule_of_the_slice_sampler():
    """Coordinate i of a transition: block slot i << 14 -> (the log-uniform's uniform from words (x, y), runiform from (z, w)); shrink attempt a >= 1: block slot
    (i << 14) | ((a + 1) >> 1), words (x, y) for odd a and (z, w) for even a — two attempts per block.  Restated in Python against the oracle's draws; no
    block serves two purposes, and the slots stay inside the 24-bit field for every coordinate the layouts allow (i < 1,024) and every attempt (a < 2^14)."""
    import ctypes as C
    lib = O.load()
    seed, chain, t = 20260927, 77, 4242
    u52 = lambda hi, lo: ((int(hi) << 20 | int(lo) >> 12) + 0.5) * 2.0 ** -52
    for i in (0, 1, 99, 511, 1023):
        n = 9
        out = np.zeros(2 + n)
        lib.ko_slice_draws(C.c_uint64(seed), C.c_uint64(chain), C.c_uint64(t), i, n, out.ctypes.data)
        slots = [i << 14] + [(i << 14) | ((a + 1) >> 1) for a in range(1, n + 1)]
        assert max(slots) < 1 << 24 and ((i << 14) | ((16383 + 1) >> 1)) < ((i + 1) << 14)
        blocks = dict(zip(sorted(set(slots)), O.stream_blocks(seed, chain, t, sorted(set(slots)))))
        b0 = blocks[i << 14]
        assert out[0] == u52(b0[0], b0[1]) and out[1] == u52(b0[2], b0[3])
        for a in range(1, n + 1):
            b = blocks[(i << 14) | ((a + 1) >> 1)]
            assert out[1 + a] == (u52(b[0], b[1]) if a & 1 else u52(b[2], b[3])), (i, a)


def test_normal_tail_mass_on_the_host():
    """Tail mass of the generator as the kernels call it (stream blocks of consecutive chains / transitions): 4 x 10^7 draws,
    counts of |z| > 1, 2, 3, 4 within 4.5 binomial standard deviations of the normal law, second and fourth moments 1 and 3."""
    lib = O.load()
    thr = np.array([1.0, 2.0, 3.0, 4.0]); cnt = np.zeros(4, np.uint64); mom = np.zeros(4)
    nch, nt = 10000, 1000
    lib.ko_normal_tail(987654321, 5, nch, nt, 4, thr.ctypes.data, cnt.ctypes.data, mom.ctypes.data)
    ndraw = 4 * nch * nt                                      # both pairs of every block
    p = 2 * stats.norm.sf(thr)
    assert np.all(np.abs(cnt.astype(float) - ndraw * p) < 4.5 * np.sqrt(ndraw * p * (1 - p))), (cnt, ndraw * p)
    assert abs(mom[0] / ndraw) < 4.5 / math.sqrt(ndraw)
    assert abs(mom[1] / ndraw - 1.0) < 4.5 * math.sqrt(2.0 / ndraw) and abs(mom[2] / ndraw - 3.0) < 4.5 * math.sqrt(96.0 / ndraw)
    assert mom[3] < 7.9


def test_tuner_score_kats(oracle):
    # test/common.jl:6 (runs) and test/AcceptanceRateMCTuner.jl:8-14 (stale file, live functions)
    assert oracle.ko_logistic(0.7, 3, 4, 2.1, 1.4) == 1.4110527196983078
    assert oracle.ko_logistic_rate_score(0.25, 7.0) == 1.7039056039366212
    assert oracle.ko_logistic_rate_score(0.5, 11.0) == 1.991859724568208
    # kd_erf is msun's s_erf.c (Julia's erf = openlibm's) operation for operation, with the build's kd_exp in the tail: the reference's
    # two erf_rate_score vectors bit for bit, and within 1 ulp of this platform's libm everywhere
    assert oracle.ko_erf_rate_score(-0.1, 3.0) == 0.6713732405408726
    assert oracle.ko_erf_rate_score(0.93, 2.0) == 1.9914724883356396
    x = np.concatenate([np.linspace(-7, 7, 100001), np.random.default_rng(0).uniform(-3, 3, 100000), np.random.default_rng(1).uniform(-1e-3, 1e-3, 1000)])
    r, e = O.math_op(6, x), np.array([math.erf(v) for v in x])          # (libm's erf descends from the same Sun code)
    nz = e != 0
    assert np.max(np.abs(r[nz] - e[nz]) / np.spacing(np.abs(e[nz]))) <= 1
    assert np.mean(r == e) > 0.9
    import mpmath
    mpmath.mp.prec = 120
    for v, got in zip(x[::997], r[::997]):                               # and the true value is within 1 ulp
        t = mpmath.erf(mpmath.mpf(float(v)))
        assert abs(mpmath.mpf(float(got)) - t) <= mpmath.mpf(float(np.spacing(abs(float(got)) or 5e-324)))
    assert O.math_op(6, [0.0])[0] == 0.0 and O.math_op(6, [np.inf])[0] == 1.0 and O.math_op(6, [-9.0])[0] == -1.0


def _diag_job(mu, sigma):
    import klara_jl_amd as K
    t = K.GaussDiagTarget.mvnormal(mu, sigma)
    return O.OracleJob(sampler=L.SAMPLER_MALA, target_kind=t.kind, nchains=1, ndims=t.ndims, nsteps=1,
                       gauss_w=t.w, gauss_mu=t.mu, gauss_const=t.const)


@pytest.mark.parametrize("x,mu,sigma", [
    ([5.18, -7.76], [6.11, -8.5], 1.0),          # test/BasicContMuvParameter.jl:39-51
    ([-11.87, -13.44], [-20.2, -18.91], 1.0),    # :62-75
    ([3.79, 4.64], [5.4, 5.3], 1.0),             # :143-156
    ([-1.91, -0.9], [0.12, 0.99], 1.0),          # :167-180
    ([1.25, 1.8], [0.0, 0.0], [10.0, 2.0]),      # :88-100 diagonal prior
    ([-0.21, 0.98], [0.0, 0.0], [1.0, 1.0]),     # :114-127
])
def test_mvnormal_target_closures(x, mu, sigma):
    """logtarget!/gradlogtarget! synthesised from MvNormal: lt == logpdf, glt == gradlogpdf."""
    lt, g = _diag_job(mu, sigma).eval_target(x)
    sg = np.broadcast_to(np.asarray(sigma, float), (2,))
    assert lt == pytest.approx(stats.multivariate_normal(mu, np.diag(sg ** 2)).logpdf(x), rel=1e-13, abs=1e-13)
    assert np.allclose(g, -(np.asarray(x) - np.asarray(mu)) / sg ** 2, rtol=1e-14, atol=0)


def test_mvnormal_first_kat_value():
    lt, g = _diag_job([6.11, -8.5], 1.0).eval_target([5.18, -7.76])
    assert lt == pytest.approx(-2.544127066409346, rel=1e-14)
    assert np.allclose(g, [0.93, -0.74], rtol=1e-13)


def test_unnormalised_target_closure():
    # test/BasicContMuvParameter.jl:539-563: lt = -(x-mu).(x-mu), glt = -2(x-mu)
    import klara_jl_amd as K
    x, mu = np.array([-4.29, 2.91]), np.array([2.2, 2.02])
    t = K.GaussDiagTarget(2, mu=mu)
    job = O.OracleJob(sampler=L.SAMPLER_MALA, target_kind=t.kind, nchains=1, ndims=2, nsteps=1, gauss_mu=mu)
    lt, g = job.eval_target(x)
    ref = stats.multivariate_normal(mu, np.eye(2)).logpdf(x)
    assert 0.5 * (lt - 2 * math.log(2 * math.pi)) == pytest.approx(ref, rel=1e-12)
    assert np.allclose(0.5 * g, -(x - mu))


def test_dense_target_with_mean_matches_scipy():
    # the builder-defined dense target with a mean: lt = c - 1/2 (x-mu)' P (x-mu), grad = -P (x-mu) — against scipy's MvNormal log-density
    rng = np.random.default_rng(4)
    d = 23
    a = rng.standard_normal((d, d)); cov = a @ a.T / d + np.eye(d); prec = np.linalg.inv(cov)
    mu, x = rng.standard_normal(d) * 3, rng.standard_normal(d) * 2
    c = -0.5 * (d * math.log(2 * math.pi) + np.linalg.slogdet(cov)[1])
    job = O.OracleJob(sampler=L.SAMPLER_HMC, target_kind=L.TARGET_GAUSS_DENSE, nchains=1, ndims=d, nsteps=1, gauss_prec=prec, gauss_mu=mu,
                      gauss_const=c)
    lt, g = job.eval_target(x)
    assert lt == pytest.approx(stats.multivariate_normal(mu, cov).logpdf(x), rel=1e-11)
    assert np.allclose(g, -prec @ (x - mu), rtol=1e-11, atol=1e-12)


def test_logistic_target_matches_numpy():
    X, y = cases.swiss_data()
    p = np.array([5.1, -0.9, 8.2, -4.5])
    job = O.OracleJob(sampler=L.SAMPLER_MALA, target_kind=L.TARGET_LOGISTIC, nchains=1, ndims=4, nsteps=1,
                      logit_X=X, logit_y=y, logit_lambda=100.0)
    lt, g = job.eval_target(p)
    xp = X @ p
    ref_lt = xp @ y - np.sum(np.log(1 + np.exp(xp))) - 0.5 * (p @ p / 100.0 + 4 * np.log(2 * np.pi * 100.0))
    ref_g = X.T @ (y - 1 / (1 + np.exp(-xp))) - p / 100.0
    assert lt == pytest.approx(ref_lt, rel=1e-12)
    assert np.allclose(g, ref_g, rtol=1e-11)


def test_logistic_target_with_20_parameters_matches_numpy():
    # beyond 16 parameters: on the matrix cores since round 6 (layout kind 5: row sums over 4 lane-quarters, klara_logit_mfma.h); KLARA_LOGIT_NO_MFMA=1 gives
    # the run-time compiled closure form of rounds 1-5 (all rows on one lane).  The same closures either way, and the same fma chains for X p and X' r.
    X, y = cases.synthetic_logit(300, 20, seed=3)
    p = 0.3 * np.random.default_rng(1).standard_normal(20)
    got = {}
    os.environ.pop("KLARA_LOGIT_NO_MFMA", None)
    for sampler, lay in ((L.SAMPLER_MALA, (5, 4, 8)), (L.SAMPLER_SLICE, (0, 1, 32))):
        if sampler == L.SAMPLER_SLICE:
            os.environ["KLARA_LOGIT_NO_MFMA"] = "1"          # the closure form (every sampler of a kind-5 job is on the matrix cores; this is the library's A/B switch)
        job = O.OracleJob(sampler=sampler, target_kind=L.TARGET_LOGISTIC, nchains=1, ndims=20, nsteps=1, logit_X=X, logit_y=y, logit_lambda=10.0,
                          slice_widths=np.ones(20) if sampler == L.SAMPLER_SLICE else None)
        assert (job.layout.kind, job.layout.G, job.layout.E) == lay
        lt, g = job.eval_target(p)
        xp = X @ p
        assert lt == pytest.approx(xp @ y - np.sum(np.log(1 + np.exp(xp))) - 0.5 * (p @ p / 10.0 + 20 * np.log(2 * np.pi * 10.0)), rel=1e-12)
        assert np.allclose(g, X.T @ (y - 1 / (1 + np.exp(-xp))) - p / 10.0, rtol=1e-11, atol=1e-12)
        got[sampler] = (lt, g)
    os.environ.pop("KLARA_LOGIT_NO_MFMA", None)
    assert np.array_equal(got[L.SAMPLER_MALA][1], got[L.SAMPLER_SLICE][1])         # the gradient chains are the same in both layouts; the log-target's sums are not
    assert got[L.SAMPLER_MALA][0] == pytest.approx(got[L.SAMPLER_SLICE][0], rel=1e-13)
    # 128 parameters, 33 rows (a third tile with one row): the layout's largest NE
    X, y = cases.synthetic_logit(33, 128, seed=4)
    p = 0.1 * np.random.default_rng(2).standard_normal(128)
    job = O.OracleJob(sampler=L.SAMPLER_HMC, target_kind=L.TARGET_LOGISTIC, nchains=1, ndims=128, nsteps=1, logit_X=X, logit_y=y, logit_lambda=10.0)
    assert (job.layout.kind, job.layout.G, job.layout.E) == (5, 4, 32)
    lt, g = job.eval_target(p)
    xp = X @ p
    assert lt == pytest.approx(xp @ y - np.sum(np.log(1 + np.exp(xp))) - 0.5 * (p @ p / 10.0 + 128 * np.log(2 * np.pi * 10.0)), rel=1e-12)
    assert np.allclose(g, X.T @ (y - 1 / (1 + np.exp(-xp))) - p / 10.0, rtol=1e-11, atol=1e-12)


def test_hierarchical_rats_target_matches_numpy_and_finite_differences():
    """Builder-defined BUGS 'Rats' target on the reference's data files (data/rats/*.csv): closed form in
    NumPy for lt, central differences for the gradient."""
    t = cases.rats_target()
    th = t.least_squares_start() + 0.01 * np.random.default_rng(0).standard_normal(t.ndims)
    c = cases.make_case("hmc_rats")
    lt, g = O.OracleJob(**cases.oracle_kwargs(c)).eval_target(th)

    def lt_np(v):
        a, b = v[0:60:2], v[1:60:2]
        ac, bc, sc, sa, sb = v[60:]
        r = t.Y - a[:, None] - b[:, None] * t.xc[None, :]
        wc, wa, wb = np.exp(-2 * sc), np.exp(-2 * sa), np.exp(-2 * sb)
        return (-150 * sc - 0.5 * wc * (r ** 2).sum() - 30 * sa - 0.5 * wa * ((a - ac) ** 2).sum()
                - 30 * sb - 0.5 * wb * ((b - bc) ** 2).sum() - 0.5 * t.prior_prec * (ac ** 2 + bc ** 2)
                + sum(-2 * t.gamma_a * s - t.gamma_b * np.exp(-2 * s) for s in (sc, sa, sb)))

    assert lt == pytest.approx(lt_np(th), rel=1e-13)
    for i in range(th.size):
        h = 1e-6 * max(1.0, abs(th[i]))
        e = np.zeros_like(th); e[i] = h
        fd = (lt_np(th + e) - lt_np(th - e)) / (2 * h)
        assert g[i] == pytest.approx(fd, rel=2e-5, abs=1e-5), i


@pytest.mark.parametrize("mu,x,s,mu0,s0", [
    ([-2.637, -1.132], [-1.88, 2.23], [1.0, 1.0], [0.0, 0.0], [1.0, 1.0]),        # test/BasicContMuvParameter.jl:235-281
    ([5.59, -7.25], [4.11, 8.17], [1.0, 1.0], [0.0, 0.0], [1.0, 1.0]),            # :328-396
    ([4.21, 7.91], [-3.1, -2.52], [2.0, 1.0], [1.0, 2.5], [3.0, 5.0]),            # :402-449
    ([6.69, -3.125], [5.43, 9.783], [1.0, 1.0], [0.0, 0.0], [1.0, 1.0]),          # :458-530
])
def test_likelihood_prior_closures_normal_normal_kats(mu, x, s, mu0, s0):
    """The reference's own known answers for a parameter built from loglikelihood / logprior (and their gradient) closures:
    ll = logpdf(MvNormal(mu, S), x), lp = logpdf(MvNormal(mu0, S0), mu), lt = ll + lp, gll = S^-1 (x - mu), glp = -S0^-1 (mu - mu0),
    glt = gll + glp (`isapprox` in the reference).  Here: the closures as C text (tests/cases.py), compiled for the host exactly as
    the device compiles them, the composition supplied by klara_custom_compose.h (BasicContMuvParameter.jl:184-189); reference
    values from SciPy's multivariate normal."""
    import ctypes as C
    t = cases.normal_normal_target(x, s, mu0, s0)
    assert t.has_parts
    lib, lt_ptr, grad_ptr = O.compile_user_target(t.source, 2)[:3]
    mu = np.array(mu, float); data = t.data
    dp = C.POINTER(C.c_double)
    for f in (lib.klara_user_loglikelihood, lib.klara_user_logprior, lib.klara_user_logtarget):
        f.restype = C.c_double; f.argtypes = [dp, C.c_int, dp, C.c_longlong]
    for f in (lib.klara_user_gradloglikelihood, lib.klara_user_gradlogprior, lib.klara_user_gradlogtarget):
        f.restype = None; f.argtypes = [dp, C.c_int, dp, C.c_longlong, dp]
    a = lambda v: v.ctypes.data_as(dp)
    ll = lib.klara_user_loglikelihood(a(mu), 2, a(data), data.size); lp = lib.klara_user_logprior(a(mu), 2, a(data), data.size)
    lt = lib.klara_user_logtarget(a(mu), 2, a(data), data.size)
    gll, glp, glt = np.zeros(2), np.zeros(2), np.zeros(2)
    lib.klara_user_gradloglikelihood(a(mu), 2, a(data), data.size, a(gll)); lib.klara_user_gradlogprior(a(mu), 2, a(data), data.size, a(glp))
    lib.klara_user_gradlogtarget(a(mu), 2, a(data), data.size, a(glt))
    ref_ll = stats.multivariate_normal(mu, np.diag(s)).logpdf(x); ref_lp = stats.multivariate_normal(mu0, np.diag(s0)).logpdf(mu)
    assert math.isclose(ll, ref_ll, rel_tol=1e-13) and math.isclose(lp, ref_lp, rel_tol=1e-13)
    assert lt == ll + lp                                                     # the composition is the sum of the two parts, bit for bit
    assert np.allclose(gll, (np.array(x) - mu) / np.array(s), rtol=1e-15) and np.allclose(glp, -(mu - np.array(mu0)) / np.array(s0), rtol=1e-15)
    assert np.array_equal(glt, gll + glp)
    # and through the oracle's target evaluation (what the samplers call)
    job = O.OracleJob(sampler=L.SAMPLER_MALA, target_kind=L.TARGET_CUSTOM, nchains=1, ndims=2, nsteps=1, custom_src=t.source, custom_data=t.data)
    lt2, g2 = job.eval_target(mu)
    assert lt2 == lt and np.array_equal(g2, glt)


def test_tuner_cadence_and_counters():
    """samplers.jl:29-45: totproposed starts at period => exactly burnin/period tuning events
    (10 for 1000/100); after burn-in `proposed` keeps growing (iterate/MALA.jl:130-152)."""
    job = O.OracleJob(sampler=L.SAMPLER_MALA, target_kind=L.TARGET_GAUSS_DIAG, nchains=3, ndims=2, nsteps=1300,
                      burnin=1000, driftstep=1.0, tuner=L.TUNER_ACCEPT_RATE, targetrate=0.574, period=100)
    assert job.set_state(np.zeros((3, 2))) == 0
    assert list(job.totproposed) == [100] * 3
    steps = [job.step.copy()]
    for _ in range(13):
        job.run(100)
        steps.append(job.step.copy())
    changes = sum(int(np.any(steps[i + 1] != steps[i])) for i in range(13))
    assert changes == 10
    assert list(job.totproposed) == [1100] * 3
    assert list(job.proposed) == [300] * 3


def test_vanilla_nonverbose_never_counts():
    job = O.OracleJob(sampler=L.SAMPLER_MALA, target_kind=L.TARGET_GAUSS_DIAG, nchains=2, ndims=2, nsteps=300,
                      burnin=100, driftstep=0.5)
    job.set_state(np.ones((2, 2)))
    job.run(300)
    assert list(job.proposed) == [0, 0] and list(job.accepted) == [0, 0] and list(job.totproposed) == [100, 100]


def test_nonfinite_init_is_reported():
    job = O.OracleJob(sampler=L.SAMPLER_MALA, target_kind=L.TARGET_GAUSS_DIAG, nchains=2, ndims=2, nsteps=10,
                      driftstep=0.5)
    assert job.set_state([[0.0, 1.0], [np.inf, 0.0]]) == L.ERR_NONFINITE_INIT


def test_results_do_not_depend_on_sharding():
    """Philox key uses the GLOBAL chain id, so chains [4,8) of an 8-chain job equal
    a 4-chain job with chain_offset=4."""
    c = cases.make_case("mala_d100")
    full = O.OracleJob(**cases.oracle_kwargs(c, nchains=8)); full.init_state_normal(); full.run(20)
    part = O.OracleJob(**cases.oracle_kwargs(c, nchains=4, chain_offset=4)); part.init_state_normal(); part.run(20)
    assert np.array_equal(full.X[4:], part.X) and np.array_equal(full.accept[:, 4:], part.accept)


def test_run_is_resumable():
    c = cases.make_case("hmc_d100")
    a = O.OracleJob(**cases.oracle_kwargs(c)); a.init_state_normal(); a.run(30)
    b = O.OracleJob(**cases.oracle_kwargs(c)); b.init_state_normal(); b.run(7); b.run(23)
    assert np.array_equal(a.X, b.X) and np.array_equal(a.accept, b.accept) and np.array_equal(a.sum, b.sum)


@pytest.mark.parametrize("name", cases.GOLDEN_CASES)
def test_oracle_reproduces_golden(name):
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", cases.GOLDEN / "make_golden.py")
    mg = importlib.util.module_from_spec(spec); spec.loader.exec_module(mg)
    out = mg.run_case(name)
    gold = np.load(cases.GOLDEN / f"{name}.npz")
    for k in gold.files:
        assert np.array_equal(out[k], gold[k], equal_nan=True), (name, k)


def test_layout_changes_only_the_summation_order():
    """The lane layout a handle reports (klara_get_layout) fixes the order of the per-chain sums and nothing else: the
    same MALA job replayed by the oracle under the group layout (kind 0) and the pair-transposed layout (kind 3) draws
    the same normals, takes the same decisions (a flip would need a Metropolis ratio within an ulp of log u) and ends
    with log-targets equal to rounding."""
    case = cases.make_case("dt_mala_d100_small_step")
    a = O.OracleJob(**cases.oracle_kwargs(case, layout=(0, 32, 4)))
    b = O.OracleJob(**cases.oracle_kwargs(case, layout=(3, 8, 14)))
    for j in (a, b):
        assert j.init_state_normal() == 0
    assert np.array_equal(a.X, b.X) and np.allclose(a.LT, b.LT, rtol=1e-14)
    for j in (a, b):
        assert j.run(case["nsteps"]) == 0
    assert np.array_equal(a.accept, b.accept)
    assert np.allclose(a.X, b.X, rtol=0, atol=0) and np.allclose(a.LT, b.LT, rtol=1e-13)
    assert not np.array_equal(a.LT, b.LT)           # ...but the sums really are taken in a different order


@pytest.mark.parametrize("d", [20, 100, 257, 300, 700, 1000])
def test_layout_kind6_sums_in_the_order_of_the_even_tile_deal(d):
    """Layout kind 6 (klara_dense_split.h, round 6): the ceil(D / 16) row tiles dealt evenly to W = 4 ceil(MT / 16) wavefronts (consecutive tiles, the first MT % W one more);
    lane partials over the wavefront's elements i with i % 4 = q in ascending order, (q0 + q1) + (q2 + q3), then the wavefronts in ascending order.  The oracle's log-target
    of a dense Gaussian under that layout against this restatement in Python floats (from the oracle's own gradient: lt = c + 1/2 sum_i (x_i - mu_i) g_i)."""
    rng = np.random.default_rng(d)
    a = rng.standard_normal((d, d)); prec = a @ a.T / d + np.eye(d)
    mu = rng.standard_normal(d)
    mt = (d + 15) // 16; w = 4 * ((mt + 15) // 16)
    job = O.OracleJob(sampler=L.SAMPLER_HMC, target_kind=L.TARGET_GAUSS_DENSE, nchains=3, ndims=d, nsteps=4, leapstep=0.1, nleaps=2, gauss_prec=prec, gauss_mu=mu,
                      gauss_const=0.5, layout=(6, w, 16))
    x0 = rng.standard_normal((3, d)) + mu
    assert job.set_state(x0) == 0
    base, rem = divmod(mt, w)
    for c in range(3):
        terms = (x0[c] - mu) * job.G[c]
        tot = None
        for wv in range(w):
            t = base + (1 if wv < rem else 0); t0 = wv * base + min(wv, rem)
            pq = [0.0, 0.0, 0.0, 0.0]
            for i in range(16 * t0, min(d, 16 * (t0 + t))):
                pq[i & 3] = pq[i & 3] + float(terms[i])
            v = (pq[0] + pq[1]) + (pq[2] + pq[3])
            tot = v if tot is None else tot + v
        assert job.LT[c] == 0.5 + 0.5 * tot, (d, c)
    other = O.OracleJob(sampler=L.SAMPLER_HMC, target_kind=L.TARGET_GAUSS_DENSE, nchains=3, ndims=d, nsteps=4, leapstep=0.1, nleaps=2, gauss_prec=prec, gauss_mu=mu,
                        gauss_const=0.5, layout=(1, 4, 8 * ((d + 31) // 32)))
    assert other.set_state(x0) == 0
    assert np.array_equal(other.G, job.G) and np.allclose(other.LT, job.LT, rtol=1e-13)          # (the gradient's fma chains do not depend on the layout)


def test_posterior_moments_mh_readme():
    """BASELINE cfg 1: README MH example, truth mean 0, var 1/2 (lt = -|x|^2). 64 replicas x 10000 steps."""
    job = O.OracleJob(sampler=L.SAMPLER_MH, target_kind=L.TARGET_GAUSS_DIAG, nchains=64, ndims=2, nsteps=10000,
                      burnin=1000, mh_sigma=[1.0, 1.0], want_accept=False)
    job.set_state(np.tile([5.1, -0.9], (64, 1)))
    job.run(10000)
    n = 9000 * 64
    m = job.sum.sum(0) / n
    v = job.sumsq.sum(0) / n - m * m
    assert np.all(np.abs(m) < 0.01) and np.all(np.abs(v - 0.5) < 0.01)
    assert 0.35 < job.naccept.mean() / 10000 < 0.5


def test_posterior_moments_hmc_dense():
    d = 16
    p = cases.compound_symmetric_precision(d, 0.5)
    job = O.OracleJob(sampler=L.SAMPLER_HMC, target_kind=L.TARGET_GAUSS_DENSE, nchains=256, ndims=d, nsteps=600,
                      burnin=100, leapstep=0.3, nleaps=5, gauss_prec=p, want_accept=False)
    job.init_state_normal()
    job.run(600)
    n = 500 * 256
    m = job.sum.sum(0) / n
    v = job.sumsq.sum(0) / n - m * m
    assert np.all(np.abs(m) < 0.03) and np.all(np.abs(v - 1.0) < 0.05)


def test_slice_sampler_moments():
    job = O.OracleJob(sampler=L.SAMPLER_SLICE, target_kind=L.TARGET_GAUSS_DIAG, nchains=64, ndims=5, nsteps=1100,
                      burnin=100, slice_widths=[1.0] * 5, want_accept=False)
    job.init_state_normal()
    assert job.run(1100) == 0
    n = 1000 * 64
    m = job.sum.sum(0) / n
    v = job.sumsq.sum(0) / n - m * m
    assert np.all(np.abs(m) < 0.02) and np.all(np.abs(v - 0.5) < 0.02)


def test_hierarchical_target_accuracy_against_200_bit_arithmetic():
    """The hierarchical target forms a unit's residual sums from its sufficient statistics (sum y, sum y x, sum y^2); the
    cancellation in sum r^2 = Syy + a (T a - 2 Sy) + b (...) costs little: log-target and gradient agree with a 200-bit
    evaluation of the model as written (r_ij = Y_ij - a_i - b_i xc_j) to ~1e-14 relative around the rats posterior."""
    mp = pytest.importorskip("mpmath")
    t = cases.rats_target()
    Y, xc = t.Y, t.xc
    R, T = Y.shape
    rng = np.random.default_rng(0)
    x0 = t.least_squares_start()
    job = O.OracleJob(sampler=L.SAMPLER_HMC, target_kind=L.TARGET_HIER_NORMAL, nchains=1, ndims=t.ndims, nsteps=10, hier_Y=Y, hier_xc=xc)
    mp.mp.prec = 200
    p0, a0, b0 = mp.mpf(1e-4), mp.mpf(1e-3), mp.mpf(1e-3)
    for _ in range(12):
        th = x0 + 0.05 * rng.standard_normal(t.ndims)
        lt, g = job.eval_target(th)
        a = [mp.mpf(v) for v in th[0:2 * R:2]]; b = [mp.mpf(v) for v in th[1:2 * R:2]]
        ac, bc, sc, sa, sb = [mp.mpf(v) for v in th[2 * R:]]
        wc, wa, wb = mp.e ** (-2 * sc), mp.e ** (-2 * sa), mp.e ** (-2 * sb)
        res = [[mp.mpf(Y[i, j]) - a[i] - b[i] * mp.mpf(xc[j]) for j in range(T)] for i in range(R)]
        C2 = sum(r * r for row in res for r in row)
        A2 = sum((a[i] - ac) ** 2 for i in range(R)); B2 = sum((b[i] - bc) ** 2 for i in range(R))
        ref = ((-R * T * sc - wc * C2 / 2) + (-2 * a0 * sc - b0 * wc) + (-R * sa - wa * A2 / 2) + (-2 * a0 * sa - b0 * wa)
               + (-R * sb - wb * B2 / 2) + (-2 * a0 * sb - b0 * wb) - p0 / 2 * (ac ** 2 + bc ** 2))
        assert abs((mp.mpf(lt) - ref) / ref) < 1e-12
        for i in range(R):
            ga = wc * sum(res[i]) - wa * (a[i] - ac)
            gb = wc * sum(res[i][j] * mp.mpf(xc[j]) for j in range(T)) - wb * (b[i] - bc)
            assert abs(mp.mpf(g[2 * i]) - ga) < 1e-12 * (abs(ga) + 1) and abs(mp.mpf(g[2 * i + 1]) - gb) < 1e-12 * (abs(gb) + 1)
        gsc = (wc * C2 - R * T - 2 * a0) + 2 * b0 * wc
        assert abs(mp.mpf(g[2 * R + 2]) - gsc) < 1e-11 * (abs(gsc) + 1)
