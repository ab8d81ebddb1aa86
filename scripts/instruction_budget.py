#!/usr/bin/env python3
"""NECESSARY vector-instruction budgets of the transition kernels bench.py prices.

`roofline.frac` in the bench line is  necessary VALU issue-cycles / elapsed SIMD-cycles:  the instructions the algorithm AS SPECIFIED
needs (this file: one count per source-level operation of detmath.h and of the sampler / target arithmetic, at its cheapest gfx950
encoding), 4 issue cycles each (one FP64 / integer vector instruction of a 64-lane wavefront on a 16-lane SIMD; v_mad_u64_u32 and the
quarter-rate v_rsq_f64 / v_rcp_f64 are counted as ONE instruction like everything else, which makes the budget too small, i.e. the
fraction conservative — `extra_issue_units` below carries what they really cost), divided by launch duration x 1,024 SIMDs x 2.4 GHz.  The instructions a kernel actually ISSUES come from the
PMC summaries (SQ_INSTS_VALU per launch, profiles/r3_pmc_kernels.json) and from the compiler's assembly (scripts/asm_loops.py); the
bench line carries both and their ratio.  VALU-busy time / elapsed time is reported as `utilisation`: it says the pipe is full,
not that the kernel is tight.

What is NOT in a budget: address arithmetic, loop control, register moves, exec-mask handling, range guards the inputs of the
workload cannot trigger, instructions executed for padding lanes / padding element slots, work of rarely taken branches (the commit
of an accepted proposal, the fold of the running sums).  What IS in it although one could argue: the cross-lane data movement of the
reductions (2 v_mov_dpp per double and butterfly step), the replicated per-lane evaluation of per-chain scalars (every lane of a
chain executes them anyway: SIMD), the divisions and unfused multiply-adds the reference's formulas state.

usage: instruction_budget.py            prints the tables
"""

# ---- building blocks: operations of one evaluation on one lane (klara.jl_amd/csrc/detmath.h) -------------------------------------
PHILOX = 10 * (2 + 2) + 1        # 10 rounds x (2 v_mad_u64_u32: hi and lo of both products; 2 v_bitop3 xor3) + the per-lane counter word;
                                 # the key schedule is wave-uniform (scalar unit)
UNIT_BITS = 3                    # v_lshrrev + v_or (exponent | top 20 bits), v_alignbit (low word)
U52 = UNIT_BITS + 1              # ... and the exact subtraction (the slice sampler's 52-bit uniforms)
U44 = 3 + 1                      # the normals' radius uniform / the accept uniform: v_alignbit (exponent | top 20 bits), v_lshlrev + v_alignbit (low word), subtraction
ANGLE_BITS = 1                   # the 20-bit angle: one v_alignbit (the low word is 0)
LOG_U01 = 23                     # tmp, index (bfe), k (ashr), hz (and, sub), table offset, r (fma), dk (cvt), w (fma), hi (add), lo (sub, add, fma),
                                 # r2 (mul), 5 fma of the degree-7 polynomial, r*r2, 2 fma, final add  [the two table words: one ds_read_b128]
SQRT_RAD = 1 + 10                # -2 log u; v_rsq_f64 + 9 mul / fma (kd_sqrt_radicand)
SINCOS = 18                      # j (bfe), centre word (v_and_or), t (sub), y (2 fma), z (mul), sin y (mul + 3 fma), cos y - 1 (2 fma + mul), table offset,
                                 # rotation (4 fma)
BOX_MULLER = U44 + ANGLE_BITS + LOG_U01 + SQRT_RAD + SINCOS + 2        # 64 bits -> two normals (kd_normal_pair_w): ... + the two products radius x (cos, sin)  = 59
NORMAL_PAIR = PHILOX / 2 + BOX_MULLER                                   # a pair whose block also serves the pair 8 further on (round 4): 79.5  (rounds 1-3: a block per pair, 102)
NORMAL_PAIR_OWN = PHILOX + BOX_MULLER                                   # a pair with a block of its own (fewer than 9 pairs, or the partner pair sits in another lane): 100


def philox_blocks(npairs: int) -> int:
    """blocks of one transition's normals of one chain: pairs p and p + 8 share block (p & 7) + 8 (p >> 4)"""
    return 8 * (npairs // 16) + min(npairs % 16, 8)


EXP = 39                         # kd_exp: 2 clamps, rounding offset (bfi) + mul + add, 2 cvt, r (2 fma), index / exponent (and, ashr), table offset, r2, r4,
                                 # 4 fma, y (fma, add), e/2 and e - e/2 (3), two scale words (2 x (add, shl)), 2 mul, three special-case selects (3 x 3)
EXP_NEG = 18                     # kd_exp_neg (round 4: k from the low mantissa bits of one fma): max, fma, sub, r (2 fma), table offset (and, shl), r2, r4, 4 fma,
                                 # y (fma, add), exponent word (shl, and, add)
LOG12 = 12                       # kd_log12 on [1, 2]: bin (sub, shr, min), table offset, r (fma), 5 fma of the polynomial, r2, logc + r, final fma  [one ds_read_b128]
DIV = 10                         # IEEE f64 division as hipcc emits it: 2 v_div_scale, v_rcp, 5 fma, v_div_fmas, v_div_fixup
DIV_UNIT = 8                     # kd_div_unit_range (numerator in [2^-1021, 1] or 0, denominator in [1, 2]): v_rcp, 2 x 2 fma, mul, 2 fma
BFLY = 3                         # one butterfly step of one double: 2 v_mov_dpp (or ds_bpermute) + v_add_f64


def mala_diag_unitw_element():
    """iterate/MALA.jl:83-92 on one element of lt = -|x|^2 (no contraction: Julia evaluates a*b + c in two roundings):
    grad (1), drift mean (mul, add), proposal (mul, add), its square (1), its gradient (1), sum (1), q1 (sub, mul, mul, add), backward mean
    (mul, add), q2 (sub, mul, mul, add)"""
    return 1 + 2 + 2 + 1 + 1 + 1 + 4 + 2 + 4          # 18


def headline(lanes_per_chain: int = 4, ndims: int = 100):
    """k_diagt<MALA, ..., UNITW, MON>: necessary instructions per WAVEFRONT and transition.  A chain's ceil(D/2) element pairs are one
    Philox block + one Box-Muller evaluation + two elements of sampler arithmetic each; a wavefront carries 64 / lanes chains."""
    chains = 64 // lanes_per_chain
    npairs = (ndims + 1) // 2
    pair = BOX_MULLER + 2 * mala_diag_unitw_element()                  # 59 + 36 = 95
    pairs = npairs * chains / 64.0                                     # pair evaluations per lane: 12.5 (4 lanes), 6.25 (8 lanes)
    blocks = philox_blocks(npairs) * chains / 64.0                     # Philox blocks per lane: 6.5 (4 lanes), 3.25 (8 lanes) — 26 per chain at D = 100
    # the three sums of the Metropolis ratio in the 8-lane order: 8 lanes: 3 butterfly steps; 4 lanes: two partial sums per lane, 2 steps
    # on both, then their sum
    red = 3 * 3 * BFLY if lanes_per_chain == 8 else 3 * (2 * 2 * BFLY + 1)
    accept = 1 + 3 + 2 + 2 + 1                                          # lt', ratio (3 adds), two compares, broadcast of log u, or
    book = 3                                                            # accept count, held += 1, save-rule phase
    return {"per_pair": pair, "pair_evaluations_per_lane": pairs, "philox_blocks_per_lane": blocks, "reductions": red, "accept_test": accept, "bookkeeping": book,
            "per_wave_transition": pairs * pair + blocks * PHILOX + red + accept + book, "chains_per_wave": chains}


def cfg5_hier(nleaps: int = 32, units_per_lane: int = 4):
    """k_hiert<HMC>: per wavefront (8 chains x 8 lanes), merged leapfrog (klara_hiert.h)."""
    vals = 2 * units_per_lane + 5                                      # a lane's values: (a, b) of its units + its copy of the 5 hyper-parameters
    unit = 2 + 2 + 2 + 1 + 3 + 2 + 2 + 2 + 5                           # da, db; S1; Sx; u; v (mul + 2 fma); S2; g_a (mul, fma); g_b; 5 accumulations
    exp3 = 4 + 1 + EXP + 6                                             # pick s_c / s_a / s_b by lane (2 selects), -2 s, exp, three broadcasts
    hyper = 2 + 2 + 3 * 3                                              # gradient of a_c, b_c (mul, fma) and of the three log-sigmas (fma, sub, fma)
    leap = 2 * vals + units_per_lane * unit + 5 * 3 * BFLY + exp3 + hyper      # drift + kick (one fma per value each), units, 5-value butterfly
    normals = (units_per_lane + 1) * NORMAL_PAIR_OWN + 7 * 2            # momentum: one block per unit (a lane's units are consecutive pairs: no partner 8 further on); the hyper block's three blocks and the accept
    # draw's block are spread over the four lanes of a quad (one each) and exchanged: 7 doubles x 2 v_mov_dpp
    lt_eval = units_per_lane * (unit - 4) + 5 * 3 * BFLY + exp3 + 25    # log-target of the proposal (no gradient terms), its closing arithmetic
    energy = 2 * (2 * vals + 3 * BFLY)                                  # sum p^2 before and after
    accept = EXP + 8
    per_tr = nleaps * leap + normals + lt_eval + energy + accept + vals  # + the opening half-kick
    return {"per_leapfrog": leap, "per_unit": unit, "exp_and_broadcasts": exp3, "butterfly": 5 * 3 * BFLY, "hyper_gradient": hyper,
            "kick_and_drift": 2 * vals, "normals": normals, "per_wave_transition": per_tr, "chains_per_wave": 8, "nleaps": nleaps}


def cfg4_logistic(ndata: int = 200, ndims: int = 4, rowsplit: int = 4):
    """k_transitions<MALA, LOGISTIC, E=4> with the 4-lane row split: per wavefront (16 chains) and transition."""
    row = ndims + EXP_NEG + 1 + LOG12 + 2 + 3 + DIV_UNIT + 2 + 1 + 1 + ndims + 1     # Xp; exp(-|Xp|); 1 + t; log; softplus (max, add); numerator select;
    # division; Xp*y (mul, add); sum of softplus; residual; gradient accumulations; row offset
    rows = ndata // rowsplit
    bfly = (ndims + 2) * {4: 2, 8: 3}[rowsplit] * BFLY                  # (D + 2)-value butterfly over the chain's lanes
    prior = 2 * ndims + (DIV + 2) + ndims * (DIV + 1)                   # p.p; -(p.p / lambda + const)/2; -p / lambda per component (the example's divisions)
    normals = NORMAL_PAIR_OWN + 2 * ndims + 4                           # every lane of a chain holds the whole vector; the (D + 1) / 2 blocks of the normals
    # and the accept draw's block are spread over the four lanes of a quad (one each) and exchanged (2 v_mov_dpp per normal, 2 + 2 ds_bpermute
    # for the draw)
    sampler = ndims * mala_diag_unitw_element() - ndims * 3             # MALA arithmetic per element (the target's own terms are above)
    accept = 12
    return {"per_row": row, "rows_per_lane": rows, "butterfly": bfly, "prior": prior, "normals": normals, "sampler": sampler,
            "per_wave_transition": rows * row + bfly + prior + normals + sampler + accept, "chains_per_wave": 64 // rowsplit}


def cfg1_replicas():
    """k_transitions<MH, DIAG, E=2, one chain per lane> — the README job (README.md:23-47: MH, sigma = (1, 1), lt = -dot(z, z), D = 2) as
    replicas with running sums: per wavefront (64 chains) and transition."""
    normals = NORMAL_PAIR_OWN                                           # one block = the chain's two normals (its second half has no taker at D = 2)
    proposal = 2 * 2                                                    # x + sigma z (mul, add) per element, iterate/MH.jl:79
    target = 2 * 2 + 1                                                  # -dot(z, z): mul, add per element; the sign
    ratio = 1
    # accept iff ratio > 0 or ratio > log(rand()) (MH.jl:97): the draw is made by the lanes with ratio <= 0 — about half of them, so a
    # wavefront of 64 chains always makes it
    accept = 1 + (PHILOX + U44 + LOG_U01) + 1 + 1
    commit = 2 * 2 + 2                                                  # v_cndmask per dword of x (2 doubles) and lt
    sums = 1 + (1 + 2 * (2 + 3)) + 1                                    # held += 1; fold of the state being left: cvt, sum += h x, sumsq += h (x x); held = 0
    book = 2                                                            # accept counter, save-rule phase
    return {"normals": normals, "proposal": proposal, "target": target, "accept_test": ratio + accept, "commit": commit, "running_sums": sums,
            "bookkeeping": book, "per_wave_transition": normals + proposal + target + ratio + accept + commit + sums + book, "chains_per_wave": 64}


def hmc_iso(nleaps: int = 10, ndims: int = 100, lanes_per_chain: int = 8):
    """k_diagt<HMC, NP, Q=8, fused, UNITW> on the README target (north_star's "100-dim Gaussian HMC"), no monitor: per wavefront (8 chains)
    and transition, merged fma leapfrog."""
    chains = 64 // lanes_per_chain
    pairs = (ndims + 1) // 2 * chains / 64.0                            # 6.25 pair evaluations per lane
    npairs = (ndims + 1) // 2
    blocks = philox_blocks(npairs) * chains / 64.0                      # 3.25 Philox blocks per lane
    per_pair = (BOX_MULLER                                              # the two momenta (their block: `blocks` below)
                + 2 * (2 + 2)                                           # p.p before and after (mul, add per element each)
                + 2 * 1                                                 # opening half kick (fma per element)
                + nleaps * 2 * 3                                        # per leapfrog and element: drift (fma), gradient -2x (mul), kick (fma)
                + 2 * 2)                                                # log-target of the proposal: mul, add per element
    # K0, lt', K1: three sums over 8 lanes; 4 lanes (round 5): two partial sums per lane, 2 butterfly steps on both, then their sum (the 8-lane order)
    red = 3 * 3 * BFLY if lanes_per_chain == 8 else 3 * (2 * 2 * BFLY + 1)
    accept = 2 + 2 + EXP + 1 + 1 + 1                                    # H0, H1, ratio; exp; min; the uniform comes with the padding pair's block; compare
    commit = 2 * 2 + 2                                                  # per pair slot: selects of the value pair; lt
    return {"per_pair": per_pair, "pair_evaluations_per_lane": pairs, "philox_blocks_per_lane": blocks, "reductions": red, "accept_test": accept,
            "per_wave_transition": pairs * (per_pair + 4) + blocks * PHILOX + red + accept + 2 + 1, "chains_per_wave": chains, "nleaps": nleaps}


def slice_probe_counts(width: float = 1.0, nsamples: int = 400000, seed: int = 7):
    """Mean log-target probes of one coordinate update of the slice sampler (iterate/SliceSampler.jl:60-109, stepping out) on lt = -|x|^2 in
    stationarity: (left probes, right probes, shrink attempts) per chain, and the means of the MAXIMUM over the 8 chains of a wavefront
    (round 3's kernel: the 8 chains of a wavefront shared the loops) and over 64 coordinate updates (round 4: every lane of the wavefront updates a
    coordinate of its own; the loops run until the slowest of the 64 is done).  A simulation of the procedure itself (NumPy), seeded."""
    import numpy as np
    rng = np.random.default_rng(seed)
    n = nsamples
    x = rng.standard_normal(n) * np.sqrt(0.5); u = rng.random(n); s = np.sqrt(x * x - np.log(u))     # the slice on this coordinate: |c| < s
    r = rng.random(n)
    L = x - r * width; R = x + (1 - r) * width
    nl = np.ones(n, int); nr = np.ones(n, int)
    while True:
        go = L > -s
        if not go.any(): break
        L = np.where(go, L - width, L); nl += go
    while True:
        go = R < s
        if not go.any(): break
        R = np.where(go, R + width, R); nr += go
    ns = np.zeros(n, int); done = np.zeros(n, bool)
    while not done.all():
        c = rng.random(n) * (R - L) + L
        act = ~done; ns += act
        ok = np.abs(c) < s
        done |= act & ok
        R = np.where(act & ~ok & (c > x), c, R); L = np.where(act & ~ok & (c < x), c, L)
    m8 = lambda a: float(a[:n // 8 * 8].reshape(-1, 8).max(1).mean())
    m64 = lambda a: float(a[:n // 64 * 64].reshape(-1, 64).max(1).mean())
    blocks64 = float(np.ceil(ns[:n // 64 * 64].reshape(-1, 64).max(1) / 2.0).mean())      # Philox blocks of the attempts: two attempts share one
    return {"per_chain": (float(nl.mean()), float(nr.mean()), float(ns.mean())), "max_over_8_chains": (m8(nl), m8(nr), m8(ns)),
            "max_over_64_lanes": (m64(nl), m64(nr), m64(ns)), "shrink_blocks": (float(np.ceil(ns / 2.0).mean()), blocks64)}


SLICE_PROBES = {"per_chain": (2.128745, 2.1275425, 1.46118), "max_over_8_chains": (3.57358, 3.57234, 2.81954),
                "max_over_64_lanes": (4.73024, 4.74288, 4.47424), "shrink_blocks": (1.107035, 2.46368)}     # slice_probe_counts()


def slice_diag(ndims: int = 100, counts=None, blocks=None):
    """k_diagt<SLICE, NP=7, Q=8, UNITW> (round 4: every lane updates its own coordinates, comparisons in difference form): per wavefront (8 chains =
    64 coordinate updates at a time) and coordinate SLOT of a lane.  `counts` = (left probes, right probes, shrink attempts), `blocks` = Philox blocks of
    the attempts (two attempts share one); the algorithmic budget takes an update's own mean counts, the lockstep variant the means of the maxima over the
    64 updates that share the loops."""
    nl, nr, ns = counts if counts is not None else SLICE_PROBES["per_chain"]
    nb = blocks if blocks is not None else SLICE_PROBES["shrink_blocks"][0]
    probe = 1 + 1 + 1                                                   # the candidate's term (x x), its difference to the current term, the compare
    fixed = (PHILOX + U52 + LOG_U01 + U52                               # log(rand()), runiform: one block
             + 2 + 3                                                    # l_i = x_i - r w; r_i = x_i + (1 - r) w
             + 1                                                        # the current term
             + 4)                                                       # x_i and its term: selects of the two doubles
    expand = 1 + probe + 4                                              # step, probe, interval / difference selects
    shrink = U52 + 3 + probe + 2 + 8                                    # the attempt's uniform, candidate (sub, mul, add), probe, two more compares, selects of x', t', l_i, r_i
    per = fixed + 2 * probe + (nl - 1 + nr - 1) * expand + ns * shrink + nb * PHILOX
    slots = ndims * 8 / 64.0                                            # coordinate slots per lane: 12.5
    return {"per_probe": probe, "fixed_per_coordinate": fixed, "per_expansion": expand, "per_shrink_attempt": shrink, "philox_blocks_of_the_attempts": nb,
            "probes": {"left": nl, "right": nr, "shrink": ns}, "per_wave_slot": per, "coordinate_slots_per_lane": slots,
            "per_wave_transition": per * slots + 14 + 3 * BFLY, "chains_per_wave": 8}         # + the new state's log-target: 14 adds, one butterfly


# ---- the same budgets weighted by what an instruction costs the issue port -----------------------------------------------------------
# scripts/ubench.hip (profiles/r3_ubench_instruction_costs.txt, 8 independent chains x 4 wavefronts per SIMD): against a plain vector
# instruction (v_add_f64 / v_mul_f64 / v_fma_f64 / v_bitop3: 5.0-5.4 "cycles at 2.4 GHz" on a chip that clocks ~1.9 GHz in that loop = 4
# real cycles), v_mad_u64_u32 takes 8.27 (x 1.6) and v_rsq_f64 / v_rcp_f64 ~16 (x 3.2).  `extra_issue_units` are the instruction-equivalents
# those three add to a budget; bench.py reports the fraction with and without them.
MAD_EXTRA, QUARTER_EXTRA = 0.6, 2.2
PAIR_EXTRA = 20 * MAD_EXTRA + QUARTER_EXTRA                # 20 v_mad_u64_u32 of a Philox block, the radius' v_rsq_f64


def _with_extra(b, extra):
    b = dict(b); b["extra_issue_units_per_wave_transition"] = extra
    return b


_h4, _h8, _c5, _c4 = headline(4), headline(8), cfg5_hier(), cfg4_logistic()
def _normals_extra(b):        # 20 v_mad_u64_u32 per Philox block, one v_rsq_f64 per Box-Muller evaluation
    return b["philox_blocks_per_lane"] * 20 * MAD_EXTRA + b["pair_evaluations_per_lane"] * QUARTER_EXTRA


BUDGETS = {"headline_4lane": _with_extra(_h4, _normals_extra(_h4)),
           "headline_8lane": _with_extra(_h8, _normals_extra(_h8)),
           "cfg5": _with_extra(_c5, 5 * PAIR_EXTRA),                                        # a lane's Philox / Box-Muller evaluations
           "cfg4": _with_extra(_c4, 1 * PAIR_EXTRA + _c4["rows_per_lane"] * QUARTER_EXTRA + 5 * QUARTER_EXTRA),   # + one v_rcp_f64 per row and per prior division
           "cfg1": _with_extra(cfg1_replicas(), 2 * 20 * MAD_EXTRA + QUARTER_EXTRA),
           "hmc_iso": _with_extra(hmc_iso(), _normals_extra(hmc_iso())),
           "hmc_iso_4lane": _with_extra(hmc_iso(lanes_per_chain=4), _normals_extra(hmc_iso(lanes_per_chain=4))),
           "slice_d100": _with_extra(slice_diag(), 12.5 * (1 + SLICE_PROBES["shrink_blocks"][0]) * 20 * MAD_EXTRA),
           "slice_d100_lockstep": _with_extra(slice_diag(counts=SLICE_PROBES["max_over_64_lanes"], blocks=SLICE_PROBES["shrink_blocks"][1]),
                                              12.5 * (1 + SLICE_PROBES["shrink_blocks"][1]) * 20 * MAD_EXTRA)}

if __name__ == "__main__":
    print(f"building blocks: Philox4x32-10 {PHILOX}, u52 {U52}, log(u) {LOG_U01}, radius {SQRT_RAD}, sin/cos {SINCOS}, Box-Muller on 64 bits {BOX_MULLER}, normal pair {NORMAL_PAIR} (own block: {NORMAL_PAIR_OWN}), "
          f"exp {EXP}, exp(-a) {EXP_NEG}, division {DIV}, butterfly step {BFLY}")
    for name, b in BUDGETS.items():
        print(name, {k: (round(v, 1) if isinstance(v, float) else v) for k, v in b.items()},
              f"-> {b['per_wave_transition'] / b['chains_per_wave']:.1f} per chain and transition")
