"""The slice sampler's free-running machine (klara_kernels.h slice_free_machine: the form the group-layout, dense and streamed dense kernels run since round 5) against
the serial procedure of iterate/SliceSampler.jl:60-109 — as host logic, without a GPU.

The device tests compare the kernels with the oracle bit for bit; this file pins the DESIGN of the machine: a statement-by-statement Python transcription of
slice_free_machine, run for a tile of chains that share every pass (one probe per chain and pass, whichever its coordinate and stage ask for), must take exactly
the probes the serial procedure takes for each chain — same candidates in the same order, same final state, same stuck verdicts — whatever the other chains of the
tile are doing, and needs max-over-chains(total probes) passes.  Draws are addressed by (coordinate, attempt) as on the device, so both sides read the same numbers."""
import math

import numpy as np
import pytest

MAX_ATT = 40            # (KLARA_SLICE_MAX_ATT is 16383 on the device; the guard logic is the same at any value)


def _u(seed, chain, i, slot, word):
    """a uniform in (0, 1) addressed like the device's Philox words: (chain, coordinate, block slot, word) -> number"""
    h = hash((seed, chain, i, slot, word)) & 0xFFFFFFFFFFFF
    return (h + 0.5) / float(1 << 48)


def _start_draws(seed, chain, i):
    return math.log(_u(seed, chain, i, 0, 0)), _u(seed, chain, i, 0, 1)          # log(rand()) of :66, rand() of :71


def _attempt_uniform(seed, chain, i, a):
    return _u(seed, chain, i, (a + 1) >> 1, (a + 1) & 1)                          # attempts 2k - 1 and 2k share block k (detmath.h kd_slice_attempt_uniform)


def serial(lt, x, cur, widths, stepout, seed, chain, probes):
    """iterate/SliceSampler.jl:60-109 for one chain; `probes` receives every evaluated candidate (coordinate, value); returns (x, cur, stuck)"""
    x = x.copy()
    for i in range(len(x)):
        xi, w = x[i], widths[i]
        lg, ru = _start_draws(seed, chain, i)
        logu = lg + cur
        L, R = xi - ru * w, xi + (1.0 - ru) * w

        def at(c):
            y = x.copy(); y[i] = c
            probes.append((i, c))
            return lt(y)
        if stepout:
            guard = 0
            while at(L) > logu:
                guard += 1
                if guard > MAX_ATT:
                    return x, cur, True
                L -= w
            guard = 0
            while at(R) > logu:
                guard += 1
                if guard > MAX_ATT:
                    return x, cur, True
                R += w
        a = 1
        while True:
            if a > MAX_ATT:
                return x, cur, True
            cand = _attempt_uniform(seed, chain, i, a) * (R - L) + L
            lc = at(cand)
            if lc > logu:
                x[i] = cand; cur = lc
                break
            if cand > xi: R = cand
            elif cand < xi: L = cand
            else:
                return x, cur, True
            a += 1
    return x, cur, False


def machine(lt, X, CUR, widths, stepout, seed, probes):
    """slice_free_machine, transcribed: X (chains x D) is updated in place, one pass = one probe of every chain; returns (stuck flags, passes)"""
    n, D = X.shape
    i = np.zeros(n, int); ph = np.zeros(n, int)
    stuck = np.zeros(n, bool)
    active = ~stuck & (D > 0)
    Li = np.zeros(n); Ri = np.zeros(n); logu = np.zeros(n); xi = np.zeros(n); wd = np.zeros(n)
    a = np.ones(n, int); guard = np.zeros(n, int)
    passes = 0
    while active.any():
        passes += 1
        lc = np.zeros(n)
        cand = np.zeros(n)
        for c in range(n):                                      # (all chains of the tile: the "lanes")
            ic = min(i[c], D - 1)
            starting = active[c] and ph[c] == 0
            if starting:
                lg, ru = _start_draws(seed, c, ic)
                xi[c], wd[c] = X[c, ic], widths[ic]
                logu[c] = lg + CUR[c]
                Li[c] = xi[c] - ru * wd[c]; Ri[c] = xi[c] + (1.0 - ru) * wd[c]
                a[c], guard[c] = 1, 0
                ph[c] = 1 if stepout else 3
            u = _attempt_uniform(seed, c, ic, a[c])
            cand[c] = Li[c] if ph[c] == 1 else (Ri[c] if ph[c] == 2 else u * (Ri[c] - Li[c]) + Li[c])
            if active[c]:
                X[c, ic] = cand[c]                              # in place: an accepted candidate simply stays
                probes[c].append((ic, cand[c]))
            lc[c] = (lt[c] if isinstance(lt, list) else lt)(X[c])          # the pass evaluates every chain, idle or not
        for c in range(n):
            above = lc[c] > logu[c]
            out = active[c] and ph[c] != 3 and above
            guard[c] += 1 if out else 0
            over = out and guard[c] > MAX_ATT
            if out and not over and ph[c] == 1: Li[c] -= wd[c]
            if out and not over and ph[c] == 2: Ri[c] += wd[c]
            next_stage = active[c] and ph[c] != 3 and not above
            shr = active[c] and ph[c] == 3
            acc, rej = shr and above, shr and not above
            if rej and cand[c] > xi[c]: Ri[c] = cand[c]
            if rej and cand[c] < xi[c]: Li[c] = cand[c]
            nowhere = rej and not (cand[c] > xi[c]) and not (cand[c] < xi[c])
            a[c] += 1 if rej else 0
            spent = rej and a[c] > MAX_ATT
            stuck[c] = stuck[c] or over or nowhere or spent
            if acc: CUR[c] = lc[c]
            if next_stage: guard[c] = 0
            ph[c] = ph[c] + 1 if next_stage else (0 if acc else ph[c])
            i[c] += 1 if acc else 0
            active[c] = active[c] and not stuck[c] and i[c] < D
    return stuck, passes


def _quartic(y):
    return float(-np.sum(0.3 * y ** 4 + 0.5 * y ** 2) - 0.4 * np.sum(y[:-1] * y[1:]))


@pytest.mark.parametrize("stepout", [True, False])
@pytest.mark.parametrize("D,n", [(1, 3), (5, 16), (12, 7)])
def test_machine_takes_the_serial_procedures_probes(D, n, stepout):
    rng = np.random.default_rng(100 * D + n + int(stepout))
    X0 = rng.standard_normal((n, D)); widths = rng.uniform(0.2, 3.0, D)
    seed = 7 + D
    ser, ser_probes = [], []
    for c in range(n):
        pr = []
        ser.append(serial(_quartic, X0[c], _quartic(X0[c]), widths, stepout, seed, c, pr)); ser_probes.append(pr)
    X = X0.copy(); CUR = np.array([_quartic(X0[c]) for c in range(n)])
    mp = [[] for _ in range(n)]
    stuck, passes = machine(_quartic, X, CUR, widths, stepout, seed, mp)
    for c in range(n):
        xs, cs, st = ser[c]
        assert not st and not stuck[c]
        assert mp[c] == ser_probes[c], c                        # the same candidates in the same order
        assert np.array_equal(X[c], xs) and CUR[c] == cs
    assert passes == max(len(p) for p in ser_probes)            # a pass per probe of the slowest chain: nobody waits at a stage
    lockstep = sum(max(sum(1 for (j, _) in ser_probes[c] if j == i) for c in range(n)) for i in range(D))
    assert passes <= lockstep                                   # (per-coordinate lockstep needs at least the sum over coordinates of the slowest chain's probes)


def test_machine_stuck_verdicts_match():
    """Zero width with step-out: the ends never leave the slice (guard) — every chain stops at that coordinate.  And a tile where SOME chains' targets are NaN
    anywhere but at the start of coordinate 1 (the shrink loop runs out of attempts): both procedures must say "stuck" for the same chains, and the other
    chains of the tile must end exactly where the serial procedure puts them."""
    D, n, seed = 3, 6, 11
    rng = np.random.default_rng(5)
    X0 = rng.standard_normal((n, D))
    ser = [serial(_quartic, X0[c], _quartic(X0[c]), np.array([1.0, 0.0, 1.0]), True, seed, c, []) for c in range(n)]
    X = X0.copy(); CUR = np.array([_quartic(X0[c]) for c in range(n)])
    stuck, _ = machine(_quartic, X, CUR, np.array([1.0, 0.0, 1.0]), True, seed, [[] for _ in range(n)])
    assert all(s[2] for s in ser) and stuck.all()

    def broken(c):
        x1 = X0[c, 1]
        return lambda y: _quartic(y) if y[1] == x1 else float("nan")
    lts = [broken(c) if c % 2 else _quartic for c in range(n)]
    widths = np.array([1.0, 1.0, 1.0])
    for stepout in (False, True):
        ser = [serial(lts[c], X0[c], lts[c](X0[c]), widths, stepout, seed, c, []) for c in range(n)]
        X = X0.copy(); CUR = np.array([lts[c](X0[c]) for c in range(n)])
        stuck, _ = machine(lts, X, CUR, widths, stepout, seed, [[] for _ in range(n)])
        assert [s[2] for s in ser] == list(stuck) == [bool(c % 2) for c in range(n)]
        for c in range(n):
            if not stuck[c]:
                assert np.array_equal(X[c], ser[c][0]) and CUR[c] == ser[c][1]


def _block_machine_attempts(fail_upto, max_att):
    """k_diagt_slice_free's attempt accounting (klara_diagt_slice.h), transcribed: a machine takes the shrink attempts of an update two per iteration from
    block k (attempts 2k + 1, 2k + 2; k = 0 when the update starts), the second of the LAST block (k = (max_att - 1) / 2: attempt max_att + 1) masked; it is
    stuck when the block after the last would be needed.  Attempts 1 .. fail_upto fail, the next succeeds.  Returns (attempts made, stuck)."""
    lastk = (max_att - 1) >> 1
    k, made = 0, 0
    while True:
        a1, a2 = 2 * k + 1, 2 * k + 2
        made = a1
        in1 = a1 > fail_upto
        in2 = (a2 > fail_upto) and k != lastk
        if not in1 and k != lastk:
            made = a2
        done = in1 or in2
        if done:
            return made, False
        knext = k + 1
        if knext > lastk:
            return made, True
        k = knext


def _serial_attempts(fail_upto, max_att):
    """the serial procedure's shrink loop (SliceSampler.jl:91-106 with the build's cap): attempts a = 1, 2, ... while a <= max_att"""
    a = 1
    while True:
        if a > max_att:
            return a - 1, True
        if a > fail_upto:
            return a, False
        a += 1


@pytest.mark.parametrize("max_att", [7, 15, 16383])
def test_two_attempt_blocks_make_exactly_the_serial_procedures_attempts_at_the_cap(max_att):
    """the free-running kernel declared a machine stuck after max_att - 1 failed attempts; the oracle and k_diagt<SLICE> make attempt max_att
    (slot (max_att + 1) / 2 of the 14-bit field, first half).  Same count and same verdict for every number of failing attempts around the cap."""
    for fail_upto in list(range(0, 6)) + list(range(max_att - 4, max_att + 3)):
        assert _block_machine_attempts(fail_upto, max_att) == _serial_attempts(fail_upto, max_att), (fail_upto, max_att)
