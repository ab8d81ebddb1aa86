"""Joint-law tests of the build-defined random stream.

Since ABI 5 a Philox block carries FOUR proposal normals: element pairs p and p + 8 of a transition take the two halves of block slot
(p & 7) + 8 (p >> 4); within a half, word wa and the low 12 bits of wb are the 44-bit Box-Muller radius uniform and the high 20 bits of wb
the angle (detmath.h kd_normal_pair_w / kd_normal_pair_at).  The marginal tests (tail mass on 3.4e10 draws, moments) say nothing about the
JOINT quantities the Metropolis ratios are made of, so here, on the D = 100 normals of a transition exactly as the samplers draw them:

  * sum_i z_i^2 over a transition follows chi-square(100) — Kolmogorov-Smirnov over >= 1e6 chain-transitions (GPU) / 65,536 (CPU);
  * the two pairs that share a block (p, p + 8) are independent: correlations of the values, of the squared radii, of the angles' first
    harmonics, and radius of one against angle of the other;
  * within a pair, radius bits and angle bits are independent: r^2 against cos / sin (k theta), k = 1, 2, 4; r^2 ~ Exp(1/2) and theta uniform (KS);
  * neighbouring pairs (p, p + 1: different blocks of one chain) and the same pair of neighbouring chains / transitions are uncorrelated;
  * the angle lattice has no direction on a coordinate axis: no atom of exact (to 1e-9) zeros among 1e8 normals (a true normal sample has
    none with probability 0.92; 2^20 directions that include the axes put 2^-19 of the mass there — ADVICE r4 item 1).

The CPU test runs the oracle's build of the generator (ko_transition_normals); the GPU test runs the device's
(klara_selftest_transition_normals), first checks a sample of it bit for bit against the CPU build, then applies the same statistics.
"""
import ctypes as C

import numpy as np
import pytest
from scipy import stats

import oracle_ffi as O
from klara_jl_amd import _lib as L

D = 100
SEED = 20260927


def _cpu_normals(first_chain, nchains, t, d=D, seed=SEED):
    lib = O.load()
    z = np.empty((nchains, d)); u = C.c_double(0.0)
    for c in range(nchains):
        lib.ko_transition_normals(seed, first_chain + c, t, d, z[c].ctypes.data, C.byref(u))
    return z


def joint_checks(z, nsig=4.75):
    """z: (n x 100) proposal normals, one row per chain-transition.  Returns a report; asserts every statistic."""
    n, d = z.shape
    npairs = d // 2
    rep = {"rows": n}
    # chi-square(100) of a transition's sum z^2
    s = (z * z).sum(axis=1)
    ks = stats.kstest(s, stats.chi2(d).cdf)
    rep["chi2_100_ks"], rep["chi2_100_p"] = float(ks.statistic), float(ks.pvalue)
    assert ks.pvalue > 1e-4, rep
    assert abs(s.mean() - d) < nsig * np.sqrt(2.0 * d / n) and abs(s.var() / (2.0 * d) - 1.0) < nsig * np.sqrt((2.0 + 12.0 / d) / n) * 1.5, (s.mean(), s.var())
    z0, z1 = z[:, 0:2 * npairs:2], z[:, 1:2 * npairs:2]            # (n x npairs): cosine and sine halves
    r2 = z0 * z0 + z1 * z1
    th = np.arctan2(z1, z0)
    # marginals of a pair's radius and angle
    ksr = stats.kstest(r2[:, :8].ravel(), stats.expon(scale=2.0).cdf)
    kst = stats.kstest((th[:, :8].ravel() + np.pi) / (2 * np.pi), "uniform")
    rep["r2_ks_p"], rep["theta_ks_p"] = float(ksr.pvalue), float(kst.pvalue)
    assert ksr.pvalue > 1e-4 and kst.pvalue > 1e-4, rep
    lim = nsig / np.sqrt(n)

    def corr(a, b):
        a = a - a.mean(axis=0); b = b - b.mean(axis=0)
        return (a * b).mean(axis=0) / np.sqrt((a * a).mean(axis=0) * (b * b).mean(axis=0))

    worst = 0.0
    # (a) within a pair: radius bits against angle bits
    for k in (1, 2, 4):
        for f in (np.cos, np.sin):
            worst = max(worst, float(np.max(np.abs(corr(r2, f(k * th))))))
    rep["radius_vs_angle_max_corr"] = worst
    assert worst < lim * 1.25, (rep, lim)                # (max over 6 x 50 correlations: a slightly wider band)
    # (b) the two pairs of a block: p and p + 8 for p with (p >> 3) even, both < npairs
    P = np.array([p for p in range(npairs) if (p >> 3) % 2 == 0 and p + 8 < npairs])
    Q = P + 8
    worst = 0.0
    for a, b in ((z0[:, P], z0[:, Q]), (z0[:, P], z1[:, Q]), (z1[:, P], z0[:, Q]), (z1[:, P], z1[:, Q]), (r2[:, P], r2[:, Q]),
                 (np.cos(th[:, P]), np.cos(th[:, Q])), (np.sin(th[:, P]), np.sin(th[:, Q])), (np.cos(th[:, P]), np.sin(th[:, Q])),
                 (r2[:, P], np.cos(th[:, Q])), (r2[:, P], np.sin(th[:, Q])), (np.cos(th[:, P]), r2[:, Q]), (np.sin(th[:, P]), r2[:, Q]),
                 (z0[:, P] ** 2, z0[:, Q] ** 2), (z1[:, P] ** 2, z1[:, Q] ** 2), (z0[:, P] ** 2, z1[:, Q] ** 2)):
        worst = max(worst, float(np.max(np.abs(corr(a, b)))))
    rep["block_halves_max_corr"] = worst
    assert worst < lim * 1.25, (rep, lim)
    # (c) neighbouring pairs (different blocks) and neighbouring rows (chains / transitions)
    worst = max(float(np.max(np.abs(corr(z[:, :-1], z[:, 1:])))), float(np.max(np.abs(corr(r2[:, :-1], r2[:, 1:])))),
                float(np.max(np.abs(corr(z[:-1], z[1:])))), float(np.max(np.abs(corr(z[:-1] ** 2, z[1:] ** 2)))))
    rep["neighbours_max_corr"] = worst
    assert worst < lim * 1.25, (rep, lim)
    # (d) the full correlation matrix of the 100 values and of their squares: largest off-diagonal entry
    zc = (z - z.mean(axis=0)) / z.std(axis=0)
    cm = (zc.T @ zc) / n
    np.fill_diagonal(cm, 0.0)
    q = z * z
    qc = (q - q.mean(axis=0)) / q.std(axis=0)
    cq = (qc.T @ qc) / n
    np.fill_diagonal(cq, 0.0)
    rep["max_offdiag_corr"], rep["max_offdiag_corr_of_squares"] = float(np.abs(cm).max()), float(np.abs(cq).max())
    assert np.abs(cm).max() < lim * 1.3 and np.abs(cq).max() < lim * 1.3, (rep, lim)          # (max over 4,950 entries)
    rep["near_zero"] = int((np.abs(z) < 1e-9).sum())
    return rep


def test_joint_law_of_a_transitions_normals_on_the_host_build():
    """65,536 chain-transitions x 100 normals from the CPU build of the generator (the oracle): every joint statistic above."""
    z = np.concatenate([_cpu_normals(0, 1 << 15, 3), _cpu_normals((1 << 33) + 11, 1 << 15, 123456)])
    rep = joint_checks(z)
    assert rep["near_zero"] == 0, rep                       # 6.5e6 normals (the full count is taken on the device, below)
    print(rep)


@pytest.mark.gpu
def test_joint_law_of_a_transitions_normals_on_the_device(klib, gpu_required):
    """1,048,576 chain-transitions x 100 normals from the device build (262,144 chains x 4 transitions): a sample of rows bit for bit equal to
    the CPU build, then the joint statistics, and no near-zero atom among the 1.05e8 values."""
    nch, ts = 1 << 18, (0, 1, 999, (1 << 40) - 1)            # (the last one is the initial-state stream's transition index)
    rows = []
    for t in ts:
        z = np.empty((nch, D)); u = np.empty(nch)
        L.check(klib.klara_selftest_transition_normals(0, SEED, 7, nch, t, D, z.ctypes.data, u.ctypes.data), "transition normals")
        ref = _cpu_normals(7 + nch - 64, 64, t)
        assert np.array_equal(z[-64:], ref), t
        assert 0.0 < u.min() and u.max() < 1.0 and abs(u.mean() - 0.5) < 4.75 / np.sqrt(12 * nch)
        rows.append(z)
    z = np.concatenate(rows)
    rep = joint_checks(z)
    print(rep)
    # P(|z| < 1e-9) = 8e-10 per draw: 0.08 expected among 1.05e8; an axis-aligned direction lattice would put ~400 there
    assert rep["near_zero"] <= 2, rep


# ------------------------------------------------------------------ the frozen stream's known answers (tests/golden/stream_kat.json)
def _kat():
    import json
    from pathlib import Path
    doc = json.loads((Path(__file__).resolve().parent / "golden" / "stream_kat.json").read_text())
    assert doc["abi_version"] == L.KLARA_ABI_VERSION, "the stream is frozen: the known answers change only together with KLARA_ABI_VERSION (include/klara_hip.h)"
    return doc["entries"]


def test_frozen_stream_known_answers_host_build_and_numpy_restatement():
    """The host build of detmath.h reproduces every committed value bit for bit; the independent NumPy restatement (own Philox, libm log / sqrt /
    cos / sin) reproduces the blocks and the uniforms exactly and the normals to 1e-13."""
    import numpy_mirror as M
    lib = O.load()
    for e in _kat():
        seed, chain, t = e["seed"], e["chain"], e["transition"]
        assert [int(v) for v in O.stream_blocks(seed, chain, t, [0])[0]] == e["block_slot0"] == list(M.stream_block(seed, chain, t, 0))
        for d in (100, 7):
            z = np.empty(d); u = C.c_double(0.0)
            lib.ko_transition_normals(seed, chain, t, d, z.ctypes.data, C.byref(u))
            want = np.array([float.fromhex(v) for v in e[f"normals_d{d}"]])
            assert np.array_equal(z, want), (seed, chain, t, d)
            assert u.value == float.fromhex(e[f"accept_uniform_d{d}"]) == M.accept_uniform(seed, chain, t, d)
            assert np.max(np.abs(M.normals(seed, chain, t, d) - want)) < 1e-13
        for i in (0, 99):
            out = np.empty(8)
            lib.ko_slice_draws(seed, chain, t, i, 6, out.ctypes.data)
            want = [float.fromhex(v) for v in e[f"slice_draws_coord{i}"]]
            assert list(out) == want
            b0 = M.stream_block(seed, chain, t, i << 14)
            assert [M.u52(b0[0], b0[1]), M.u52(b0[2], b0[3])] == want[:2]
            for a in range(1, 7):
                bb = M.stream_block(seed, chain, t, (i << 14) | ((a + 1) >> 1))
                assert (M.u52(bb[0], bb[1]) if a & 1 else M.u52(bb[2], bb[3])) == want[1 + a]


@pytest.mark.gpu
def test_frozen_stream_known_answers_device_build(klib, gpu_required):
    """The device build reproduces the committed normals and accept uniforms bit for bit."""
    for e in _kat():
        for d in (100, 7):
            z = np.empty((1, d)); u = np.empty(1)
            L.check(klib.klara_selftest_transition_normals(0, e["seed"], e["chain"], 1, e["transition"], d, z.ctypes.data, u.ctypes.data), "kat")
            assert np.array_equal(z[0], np.array([float.fromhex(v) for v in e[f"normals_d{d}"]])), (e["seed"], e["chain"], d)
            assert u[0] == float.fromhex(e[f"accept_uniform_d{d}"])
