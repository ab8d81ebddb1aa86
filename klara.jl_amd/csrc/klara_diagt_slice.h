// klara_diagt_slice.h — the slice sampler on the diagonal Gaussian target, pair-transposed layout (kind 3), lanes out of lockstep.
//
// iterate!(job, SliceSampler, Multivariate) (src/samplers/iterate/SliceSampler.jl:60-109) updates the D coordinates of a chain one after the
// other; on lt = c - sum_i w_i (x_i - mu_i)^2 every comparison of an update, lt(candidate) > log(rand()) + lt, reduces to
//     t_i(current) - t_i(candidate) > log(rand())                                     (difference form: DESIGN.md section 2 (8), oracle ko_slice_diag_delta)
// so coordinate i of transition t depends on NOTHING but coordinate i after transition t - 1 and its own draws (block slots (i << 14) | k of
// transition t): not on the chain's other coordinates, not on the chain's log-target.  Round 4's kernel (k_diagt<SLICE>, klara_diagt.h) used
// that to give every lane its own coordinates, but kept the 64 lanes of a wavefront in lockstep: step-out and shrink loops ran until the
// slowest of 64 updates was done (4.5 shrink attempts where an update makes 1.46; issued / necessary instructions 2.55, 0.71 scalar instructions
// per vector instruction for the votes, profiles/r4_pmc_summary.txt).
//
// Here the loops are turned inside out.  A lane takes ONE element of its chain (NM = 1, the shipped form: element Q slot + q — the chain's elements dealt to
// its lanes round robin, ceil(D / Q) slots; NM = 2: an element pair 2P, 2P + 1, two independent update machines interleaved for instruction-level
// parallelism) through ALL transitions of the launch before it moves to its next one:
//
//     for element slot of the lane:                (x, running sums, width, weight, mean of the element: registers for the whole launch)
//         until the machine(s) have made nsteps transitions:     one iteration = at most one transition of each machine
//             block A = slot base (log(rand()), runiform): what a starting machine needs
//             block B = base | (k + 1): the next two shrink attempts (k = 0 for a starting machine)
//             starting machines: slice level, interval, step-out (SliceSampler.jl:66-89)      (executed by all, kept by the starting ones)
//             two shrink attempts from B                                             (SliceSampler.jl:91-106)
//             accepted -> commit, next transition; else continue with the next attempt block in the next iteration
//
// A machine whose update needs more than two attempts (11 % of the updates on the README target) simply takes another iteration while its
// neighbours start their next transition: nobody waits for the slowest of 64, and the only wave-wide vote left per iteration is the loop's
// own.  What a wavefront still waits for is the slowest of its machines over a whole slot (nsteps transitions each): 6 % at the 128 transitions
// per launch these jobs default to.  The new state's log-target is formed once, AFTER the launch, by k_diagt_hist_lt<.., STATE> below, in the layout's
// order (lane partials over ascending elements, butterfly over the chain's Q lanes) — the same bits round 4's kernel and the oracle produce;
// nothing else depends on which lane updates which element.
//
// Scope: untuned jobs; monitors: the accept diagnostics, the running sums and the value history (any thinning / burn-in, ring or not) are lane-local
// like the updates themselves — a machine stores its element of a saved state when ITS transition ends; the log-target history of the saved states
// is formed afterwards from the saved values by k_diagt_hist_lt, in the layout's order.  A job that counts proposals (verbose tuner) runs
// k_diagt<SLICE>: the same draws, the same bits.
#pragma once
#include "klara_diagt.h"

#ifndef KLARA_DT_SLICEF_WF
#define KLARA_DT_SLICEF_WF 4      // wavefronts per SIMD the register allocator is asked to leave room for
#endif
#ifndef KLARA_DT_SLICEF_WF1
#define KLARA_DT_SLICEF_WF1 8     // ... of the one-machine form
#endif
#ifndef KLARA_DT_SLICEF_WF1S
#define KLARA_DT_SLICEF_WF1S 6    // ... with running sums or a non-unit diagonal (8 would spill the sums / weights)
#endif
#define KLARA_SLICEF_MAXNP 8

template <bool UNITW>
__device__ __forceinline__ double slicef_term(double v, double w, double m)
{
    const double dd = UNITW ? v : v - m;
    return UNITW ? dd * dd : w * (dd * dd);          // klara_diagt.h diag_elem: the same operations in the same order
}

// NM: update machines per lane (1: one element at a time, more wavefronts per SIMD; 2: an element pair, two interleaved dependency chains)
template <int Q, bool UNITW, bool SUMS, int NM>
__global__ __launch_bounds__(256, (NM == 1 ? (SUMS || !UNITW ? KLARA_DT_SLICEF_WF1S : KLARA_DT_SLICEF_WF1) : KLARA_DT_SLICEF_WF))
void k_diagt_slice_free(const KParams* __restrict__ pp, const KLaunch kl, const KAuto ka, const int NP)
{
    static_assert(NM == 1 || NM == 2, "one or two machines per lane");
    constexpr int CPW = 64 / Q;
    const KParams& p = *pp;
    // widths (and, for a non-unit diagonal, weights and means) of the 2 NP Q element slots, in dynamic LDS: sized by the job's NP at the launch
    // (klara_diagt_slice.hip) — static arrays for the largest NP were 12 KB per workgroup at 32 lanes per chain whatever the job's D
    extern __shared__ __attribute__((aligned(16))) double slicef_lds[];
    const int nslot = 2 * NP * Q;
    double* const lds_wd = slicef_lds;
    double* const lds_w = slicef_lds + nslot;
    double* const lds_mu = slicef_lds + 2 * nslot;
    const int D = p.D;
    for (int i = (int)threadIdx.x; i < nslot; i += (int)blockDim.x) {
        if (!UNITW) {
            lds_w[i] = (p.gw != nullptr && i < D) ? p.gw[i] : 1.0;
            lds_mu[i] = (p.gmu != nullptr && i < D) ? p.gmu[i] : 0.0;
        }
        lds_wd[i] = i < D ? p.vecparam[i] : 1.0;
    }
    auto_begin();
    kd_tables_to_lds();          // (ends with the workgroup barrier)
    const int lane = threadIdx.x & 63, q = lane & (Q - 1), cw = lane / Q;
    const long long wave0 = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    const long long grp = kl.group0 + wave0;
    unsigned wave_acc = 0;
    if (grp < kl.group_end && grp * CPW < p.nchains) {
        const long long first_chain = grp * CPW;
        const long long left = p.nchains - first_chain;
        const int here = left < CPW ? (int)left : CPW;
        const bool chain_ok = cw < here;
        const long long chain = first_chain + cw;
        const unsigned long long gchain = (unsigned long long)(p.chain_offset + chain);
        const __amdgpu_buffer_rsrc_t wx = group_window(p.X, first_chain, here, D);
        const bool do_sum = SUMS && p.sum != nullptr;
        const bool do_hist = SUMS && p.hist != nullptr;                  // (SUMS: a saved-sample monitor is on — running sums and / or value history)
        __amdgpu_buffer_rsrc_t wsum = wx, wsq = wx;
        int held0 = 0;                                 // (saved steps held at the current state: 0 or 1 between this sampler's launches)
        if (do_sum) {
            wsum = group_window(p.sum, first_chain, here, D); wsq = group_window(p.sumsq, first_chain, here, D);
            held0 = (int)p.held[chain_ok ? chain : 0];
        }
        const int nsteps = kl.nsteps;
        const unsigned long long seed = p.seed;
        const bool stepout = p.stepout != 0;
        const long long burnin = p.burnin, nsteps_total = p.nsteps_total;
        const int thinning = (int)p.thinning;
        gdouble* const hist0 = p.hist;
        const int hist_cols = (int)p.hist_cols;
        int held_out = held0;
        bool stuck_any = false;

        // NM = 1: the D elements of a chain are dealt to its Q lanes round robin — element ps Q + q in slot ps, ceil(D / Q) slots (13 at D = 100 on 8 lanes).
        // The coordinate updates do not care which lane makes them; dealing them by the layout's pairs (lane q: pairs q, q + Q, ...: 14 slots at D = 100, the
        // last two with work on 2 lanes of 8) was 7 % of the launch.  The new state's log-target, whose bits depend on the order of its sum, is formed in the
        // layout's order by k_diagt_hist_lt<.., STATE> right after this kernel.
#ifndef KLARA_SLICEF_PAIR_SLOTS
#define KLARA_SLICEF_PAIR_SLOTS 0     // 1: the layout's pairs (A/B builds)
#endif
        constexpr bool RR = NM == 1 && !KLARA_SLICEF_PAIR_SLOTS;
        const int nslots = RR ? (D + Q - 1) / Q : (2 / NM) * NP;
        for (int ps = 0; ps < nslots; ++ps) {
            // the lane's element(s) of this slot: NM = 2: the pair P = ps Q + q (elements 2P, 2P + 1); NM = 1: element ps Q + q
            int ei[NM]; bool eok[NM]; unsigned eoff[NM];
            double x[NM], sm[NM], sq[NM];
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                ei[m] = NM == 2 ? 2 * (ps * Q + q) + m : (RR ? ps * Q + q : 2 * ((ps >> 1) * Q + q) + (ps & 1));
                eok[m] = chain_ok && ei[m] < D;
                eoff[m] = eok[m] ? (unsigned)((cw * D + ei[m]) * 8) : KLARA_BUF_OOB;
                x[m] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(wx, eoff[m], 0, 0));      // (0 outside the window)
                sm[m] = 0.0; sq[m] = 0.0;
                if (do_sum) {
                    sm[m] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(wsum, eoff[m], 0, 0));
                    sq[m] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(wsq, eoff[m], 0, 0));
                }
            }
            double wt[NM], mu[NM], wd[NM], tcur[NM], L[NM], R[NM], lgu[NM];
            uint32_t base[NM];
            int tl[NM], k[NM], sphase[NM], scol[NM];
            int held[NM];
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                const int i = ei[m];
                const bool live = eok[m];
                wt[m] = UNITW ? 1.0 : lds_w[i]; mu[m] = UNITW ? 0.0 : lds_mu[i]; wd[m] = lds_wd[i];
                tcur[m] = slicef_term<UNITW>(x[m], wt[m], mu[m]);
                base[m] = (uint32_t)(live ? i : 0) << KLARA_SLICE_ATT_BITS;
                tl[m] = live ? 0 : nsteps; k[m] = 0; sphase[m] = kl.save_phase0; scol[m] = (int)kl.save_col0; held[m] = held0;     // (tl = nsteps + 1: the machine is stuck)
                L[m] = x[m]; R[m] = x[m]; lgu[m] = 0.0;
            }

            const uint32_t t0lo = (uint32_t)kl.t0, t0hi = (uint32_t)(kl.t0 >> 32);
            constexpr int LASTK = (KLARA_SLICE_MAX_ATT - 1) >> 1;                              // block index of attempts MAX_ATT, MAX_ATT + 1
            while (true) {
                bool act[NM], st[NM];
                kd_u32x4 A[NM], B[NM];
#pragma unroll
                for (int m = 0; m < NM; ++m) { act[m] = tl[m] < nsteps; st[m] = act[m] && k[m] == 0; }
                if (!__any(act[0] || act[NM - 1])) break;
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    // counter words of (transition << 24 | slot) for transition t0 + tl (detmath.h kd_stream_block), formed in 32-bit pieces
                    const uint32_t tlo = t0lo + (uint32_t)tl[m], thi = t0hi + (tlo < t0lo ? 1u : 0u);
                    const uint32_t c0 = tlo << 24, c1 = (tlo >> 8) | (thi << 24);
                    // A: the coordinate's block (log(rand()), runiform) — used by starting machines only; B: the block of the next two shrink attempts,
                    // slot base | (k + 1): attempts 1, 2 of a starting machine (k = 0), attempts 2k + 1, 2k + 2 of a continuing one
                    A[m] = kd_philox4x32_10(c0 | base[m], c1, (uint32_t)gchain, (uint32_t)(gchain >> 32), (uint32_t)seed, (uint32_t)(seed >> 32));
                    B[m] = kd_philox4x32_10(c0 | base[m] | (uint32_t)(k[m] + 1), c1, (uint32_t)gchain, (uint32_t)(gchain >> 32), (uint32_t)seed, (uint32_t)(seed >> 32));
                }
                // starting machines: slice level, interval, step-out.  Executed by every lane, kept by the starting ones (selects: a branch here
                // is taken in all but a few per cent of the iterations and costs the merge copies of everything it defines)
                {
                    double Ln[NM], Rn[NM], lg[NM];
#pragma unroll
                    for (int m = 0; m < NM; ++m) {
                        if (do_sum) {                                                          // the state being left is folded in first (KParams::held)
                            const bool fold = st[m] && held[m] > 0;
                            const double hf = (double)held[m];
                            const double a = sm[m] + hf * x[m], b = sq[m] + hf * (x[m] * x[m]);
                            sm[m] = fold ? a : sm[m]; sq[m] = fold ? b : sq[m];
                            held[m] = fold ? 0 : held[m];
                        }
                        lg[m] = kd_log_u01(kd_uniform_xy(A[m]));                               // :66 log(rand()); the slice level is lg + lt
                        const double ru = kd_uniform_zw(A[m]);                                 // :71
                        Ln[m] = x[m] - ru * wd[m];                                             // :72
                        Rn[m] = x[m] + (1.0 - ru) * wd[m];                                     // :73
                    }
                    if (stepout) {                                                             // :75-89
                        // Nothing but doubles is carried from trip to trip: a machine steps on a side while t(current) - t(end point) > log(rand()),
                        // re-formed from the end point at the top of every trip (a machine that is not starting compares against +inf: never).
                        double lgx[NM], dl[NM], dr[NM];
#pragma unroll
                        for (int m = 0; m < NM; ++m) {
                            lgx[m] = st[m] ? lg[m] : __builtin_inf();
                            dl[m] = tcur[m] - slicef_term<UNITW>(Ln[m], wt[m], mu[m]);
                            dr[m] = tcur[m] - slicef_term<UNITW>(Rn[m], wt[m], mu[m]);
                        }
                        // one loop exit (a second one — the guard — costs merge copies of the four end points in every trip): the trip counter is a
                        // scalar, a machine still stepping in trip n has made n - 1 steps, and whoever is still stepping when the loop ends on the
                        // guard is stuck
                        int trip = 1;
                        while (__any(dl[0] > lgx[0] || dr[0] > lgx[0] || dl[NM - 1] > lgx[NM - 1] || dr[NM - 1] > lgx[NM - 1]) && trip <= KLARA_SLICE_MAX_ATT) {
#pragma unroll
                            for (int m = 0; m < NM; ++m) {
                                // (one masked fma per side: fma(-1, wd, L) rounds like L - wd, fma(-0.0, wd, L) = L; a one-word select of the factor instead
                                // of a two-word select of the result)
                                const double fl = kd_u2d((uint64_t)(dl[m] > lgx[m] ? 0xbff00000u : 0x80000000u) << 32);
                                const double fr = kd_u2d((uint64_t)(dr[m] > lgx[m] ? 0x3ff00000u : 0x80000000u) << 32);     // (-0.0, not +0.0: x + -0.0 = x for every x, -0.0 included)
                                Ln[m] = kd_fma(fl, wd[m], Ln[m]); Rn[m] = kd_fma(fr, wd[m], Rn[m]);
                                dl[m] = tcur[m] - slicef_term<UNITW>(Ln[m], wt[m], mu[m]);
                                dr[m] = tcur[m] - slicef_term<UNITW>(Rn[m], wt[m], mu[m]);
                            }
                            trip = __builtin_amdgcn_readfirstlane(trip + 1);
                        }
#pragma unroll
                        for (int m = 0; m < NM; ++m) tl[m] = (dl[m] > lgx[m] || dr[m] > lgx[m]) ? nsteps + 1 : tl[m];      // (only possible after the guard)
                    }
#pragma unroll
                    for (int m = 0; m < NM; ++m) { L[m] = st[m] ? Ln[m] : L[m]; R[m] = st[m] ? Rn[m] : R[m]; lgu[m] = st[m] ? lg[m] : lgu[m]; }
                }
#pragma unroll
                for (int m = 0; m < NM; ++m) {                                                 // :91-106, two attempts
                    const uint32_t w0 = B[m].x, w1 = B[m].y, w2 = B[m].z, w3 = B[m].w;
                    const double c1 = kd_u52(w0, w1) * (R[m] - L[m]) + L[m];                   // :92-93
                    const double t1 = slicef_term<UNITW>(c1, wt[m], mu[m]);                    // :94
                    const bool in1 = tcur[m] - t1 > lgu[m];                                    // :95
                    const bool up1 = c1 > x[m], dn1 = c1 < x[m];
                    // (the interval only matters while the update goes on: it is narrowed as if attempt 1 had failed — when it succeeded nothing reads it
                    // again before the next transition's :72-73 replace it)
                    const double R1 = up1 ? c1 : R[m];                                         // :98
                    const double L1 = dn1 ? c1 : L[m];                                         // :100 (c1 < x excludes c1 > x)
                    const bool bad1 = !in1 && !up1 && !dn1;                                    // :102
                    const double c2 = kd_u52(w2, w3) * (R1 - L1) + L1;
                    const double t2 = slicef_term<UNITW>(c2, wt[m], mu[m]);
                    // (the last block, slot (MAX_ATT + 1) / 2, holds attempt MAX_ATT only: its second half would be attempt MAX_ATT + 1)
                    const bool in2 = tcur[m] - t2 > lgu[m] && k[m] != LASTK;
                    const bool up2 = c2 > x[m], dn2 = c2 < x[m];
                    const bool done = act[m] && (in1 || (!bad1 && in2));
                    const bool bad = act[m] && (bad1 || (!in1 && !in2 && !up2 && !dn2));
                    const bool more = act[m] && !done && !bad;
                    const int knext = st[m] ? 1 : k[m] + 1;
                    const bool full = more && knext > LASTK;                                   // (MAX_ATT attempts made — the oracle's and k_diagt<SLICE>'s count: ADVICE r5)
                    const double xn = in1 ? c1 : c2, tn = in1 ? t1 : t2;                       // :108 (the new value's term: what the next update starts from)
                    x[m] = done ? xn : x[m]; tcur[m] = done ? tn : tcur[m];
                    R[m] = up2 ? c2 : R1;
                    L[m] = dn2 ? c2 : L1;
                    if (SUMS) {                                                                // save rule: BasicMCJob.jl:226-231, BasicMCRange.jl:36
                        const uint32_t tlo = t0lo + (uint32_t)tl[m], thi = t0hi + (tlo < t0lo ? 1u : 0u);
                        const long long i1 = (long long)(((unsigned long long)thi << 32) | tlo) + 1;
                        const bool post = done && i1 > burnin && i1 <= nsteps_total;
                        const bool savenow = post && sphase[m] == 0;
                        held[m] += savenow ? 1 : 0;
                        if (do_hist && savenow && scol[m] < hist_cols)                         // copy!(nstate, state, i): this element of column scol
                            hist0[((long long)scol[m] * p.nchains + chain) * D + ei[m]] = xn;
                        scol[m] += savenow ? 1 : 0;
                        sphase[m] = post ? ((sphase[m] + 1 == thinning) ? 0 : sphase[m] + 1) : sphase[m];
                    }
                    k[m] = more ? knext : 0;
                    tl[m] = (bad || full) ? nsteps + 1 : tl[m] + (done ? 1 : 0);               // stuck: the machine stops where it is (error raised below)
                }
            }
            // the slot is done for this launch: value and sums back to memory
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(kd_uint2, x[m]), wx, eoff[m], 0, 0);
                if (do_sum) {
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(kd_uint2, sm[m]), wsum, eoff[m], 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(kd_uint2, sq[m]), wsq, eoff[m], 0, 0);
                }
            }
            if (ps == 0) held_out = held[0];
            stuck_any = stuck_any || tl[0] > nsteps || tl[NM - 1] > nsteps;
        }
        if (chain_ok && q == 0) {
            p.naccept[chain] += (unsigned long long)nsteps;                                    // (every slice transition "accepts": diagnostics are all true)
            if (do_sum) p.held[chain] = (long long)held_out;
            if (p.accept != nullptr) {
                guchar* const out = p.accept + kl.t0 * (unsigned long long)p.nchains + chain;
                for (int s = 0; s < nsteps; ++s) out[(long long)s * p.nchains] = 1;
            }
            wave_acc = (unsigned)nsteps;
        }
        if (stuck_any && chain_ok) klara_raise(p.error_flag, KLARA_ERR_SLICE_STUCK);
    }
    auto_finish(ka, wave_acc);
}

// The log-target of the saved states of a launch, formed from the saved VALUES (hist columns [col0, col0 + ncols)): lt = c - sum_i w_i (x_i - mu_i)^2 in
// the layout's order — lane partials over the lane's elements ascending, butterfly over the chain's Q lanes — i.e. the bits k_diagt<SLICE> keeps per
// saved step (its log-target after a transition is that full evaluation).  One wavefront per (column, chain group).
// STATE: the same sum over the CURRENT state X of the chain groups of the launch, into LT (ncols = 1): k_diagt_slice_free deals a chain's elements to its
// lanes without regard to the layout, so the state's log-target is formed here, after it.
template <int Q, bool UNITW, bool STATE = false>
__global__ __launch_bounds__(256) void k_diagt_hist_lt(const KParams* __restrict__ pp, const KLaunch kl, const int NP, const long long col0, const int ncols)
{
    constexpr int CPW = 64 / Q;
    const KParams& p = *pp;
    const int D = p.D;
    const int lane = threadIdx.x & 63, q = lane & (Q - 1), cw = lane / Q;
    const long long wave = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long ngroups = kl.group_end - kl.group0;
    if (wave >= ngroups * ncols) return;
    const long long grp = kl.group0 + wave % ngroups, col = col0 + wave / ngroups;
    const long long first_chain = grp * CPW;
    if (first_chain >= p.nchains || (!STATE && col >= p.hist_cols)) return;
    const long long left = p.nchains - first_chain;
    const int here = left < CPW ? (int)left : CPW;
    const __amdgpu_buffer_rsrc_t wh = STATE ? group_window(p.X, first_chain, here, D) : group_window(p.hist, col * p.nchains + first_chain, here, D);
    double red[1] = { 0.0 };
    for (int ps = 0; ps < NP; ++ps) {
        const int P = ps * Q + q;
        const unsigned off = 2 * P < D ? (unsigned)((cw * D + 2 * P) * 8) : KLARA_BUF_OOB;
        const kd_uint4 t = __builtin_amdgcn_raw_buffer_load_b128(wh, off, 0, 0);
        const double x0 = __builtin_bit_cast(double, kd_uint2{ t.x, t.y }), x1 = 2 * P + 1 < D ? __builtin_bit_cast(double, kd_uint2{ t.z, t.w }) : 0.0;
        const bool in0 = 2 * P < D, in1 = 2 * P + 1 < D;
        const double w0 = (!UNITW && p.gw != nullptr && in0) ? p.gw[2 * P] : 1.0, w1 = (!UNITW && p.gw != nullptr && in1) ? p.gw[2 * P + 1] : 1.0;
        const double m0 = (!UNITW && p.gmu != nullptr && in0) ? p.gmu[2 * P] : 0.0, m1 = (!UNITW && p.gmu != nullptr && in1) ? p.gmu[2 * P + 1] : 0.0;
        red[0] = red[0] + slicef_term<UNITW>(x0, w0, m0);
        red[0] = red[0] + slicef_term<UNITW>(x1, w1, m1);
    }
    group_allreduce<1>(red, Q, lane);
    if (cw < here && q == 0) {
        if (STATE) p.LT[first_chain + cw] = p.gconst - red[0];
        else p.hist_lt[col * p.nchains + first_chain + cw] = p.gconst - red[0];
    }
}
