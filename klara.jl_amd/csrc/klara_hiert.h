// klara_hiert.h — HMC (and MALA, MH) on the hierarchical normal target (KLARA_TARGET_HIER_NORMAL, BUGS "Rats") with few lanes per chain
// (layout kind 4).
//
// The group layout spreads a chain's D = 2R + 5 = 65 parameters over 32 lanes (17 busy) and pays the five-value
// butterfly, the hyper-parameter broadcasts and the exp(-2 s) evaluations of every gradient once per TWO chains:
// ~200 VALU instructions per chain and leapfrog step (PMC), of which the arithmetic is a fraction.  Here a chain takes
// Q = 8 lanes and a wavefront carries 8 chains: lane q owns the RPL = 4 units (rats) 4q..4q+3 — their (a_i, b_i), the
// T observations of each and the centred covariate in registers — and every lane keeps its own copy of the five
// hyper-parameters (a_c, b_c, s_c, s_a, s_b) with their momenta and gradients.  Per gradient: residual sums lane-local,
// one 5-value butterfly over 8 lanes (3 DPP steps), one exp per lane (lanes 0..2 take s_c, s_a, s_b; three broadcasts),
// the hyper-parameter gradient recomputed identically by all lanes from the reduced sums — no broadcast of the state.
// Transition arithmetic: iterate/HMC.jl:124-201 with leapfrog! samplers.jl:122-134; target: oracle ko_hier_eval
// (include/klara_hip.h gives the model).  Sums: lane partial over the lane's units ascending (for sum(p.*p) over its
// elements ascending, lane 0 then adds the five hyper terms), then the xor butterfly over the 8 lanes — oracle
// ko_reduce kind 4.
//
// Scope: HMC, MALA and MH with the Vanilla, AcceptanceRate (per chain or pooled per GPU) or (HMC) DualAveraging tuner, any monitor;
// 9 <= R <= 32 units, any number of observations per unit (they enter through sufficient statistics).  Everything else stays
// on the group layout.
#pragma once
#include "klara_kernels.h"
#include "klara_diagt.h"      // group_window / buffer helpers, kd_uint4

#define KLARA_HIERT_Q 8

template <int RPL, int NT>
struct HierLane {
    int lane, q, cw;
    int R, D;
    bool rv[RPL];              // unit RPL*q + k exists
    unsigned roff[RPL];        // byte offset of (a_i, b_i) inside the wavefront's chain window (OOB for missing units)
    unsigned hoff;             // byte offset of the hyper block (same for the 8 lanes of a chain)
    // sufficient statistics of this lane's units (sum_j y, -2 sum_j y, sum_j y x, -2 sum_j y x, sum_j y^2) and of the centred
    // covariate (sum_j x, sum_j x^2): the residual sums of an evaluation are formed from these, not from the observations
    double Sy[RPL], m2Sy[RPL], Sxy[RPL], m2Sxy[RPL], Syy[RPL], X1, X2, Td;
    double p0, a0, b0;
};

template <int RPL, int NT>
__device__ __forceinline__ HierLane<RPL, NT> make_hlane(const KParams& p)
{
    HierLane<RPL, NT> c;
    c.lane = threadIdx.x & 63;
    c.q = c.lane & (KLARA_HIERT_Q - 1);
    c.cw = c.lane / KLARA_HIERT_Q;
    c.R = p.hR; c.D = p.D;
    c.p0 = p.hp0; c.a0 = p.ha0; c.b0 = p.hb0;
    // (the number of observations per unit only matters here, once per launch: NT is no longer a code-shape parameter)
    const int T = p.hT;
    c.X1 = 0.0; c.X2 = 0.0; c.Td = (double)T;
    for (int j = 0; j < T; ++j) { const double xj = p.hxc[j]; c.X1 = c.X1 + xj; c.X2 = kd_fma(xj, xj, c.X2); }
#pragma unroll
    for (int k = 0; k < RPL; ++k) {
        const int r = RPL * c.q + k;
        c.rv[k] = r < c.R;
        c.roff[k] = c.rv[k] ? (unsigned)((c.cw * c.D + 2 * r) * 8) : KLARA_BUF_OOB;
        double sy = 0.0, sxy = 0.0, syy = 0.0;               // (a missing unit: all zero)
        for (int j = 0; j < T; ++j) {
            const double y = c.rv[k] ? p.hY[r * T + j] : 0.0;
            sy = sy + y; sxy = kd_fma(y, p.hxc[j], sxy); syy = kd_fma(y, y, syy);
        }
        c.Sy[k] = sy; c.m2Sy[k] = -2.0 * sy; c.Sxy[k] = sxy; c.m2Sxy[k] = -2.0 * sxy; c.Syy[k] = syy;
    }
    c.hoff = (unsigned)((c.cw * c.D + 2 * c.R) * 8);
    return c;
}

// state of one chain as this lane sees it: its units' (a, b) and a private copy of the hyper block
template <int RPL>
struct HierVec { double a[RPL], b[RPL], h[5]; };

template <int RPL, int NT>
__device__ __forceinline__ void hload(const HierLane<RPL, NT>& c, __amdgpu_buffer_rsrc_t w, HierVec<RPL>& v)
{
#pragma unroll
    for (int k = 0; k < RPL; ++k) {
        const kd_uint4 t = __builtin_amdgcn_raw_buffer_load_b128(w, c.roff[k], 0, 0);
        v.a[k] = __builtin_bit_cast(double, kd_uint2{ t.x, t.y });
        v.b[k] = __builtin_bit_cast(double, kd_uint2{ t.z, t.w });
    }
#pragma unroll
    for (int k = 0; k < 5; ++k)
        v.h[k] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(w, c.hoff + 8u * (unsigned)k, 0, 0));
}
template <int RPL, int NT>
__device__ __forceinline__ void hstore(const HierLane<RPL, NT>& c, __amdgpu_buffer_rsrc_t w, const HierVec<RPL>& v)
{
#pragma unroll
    for (int k = 0; k < RPL; ++k) {
        const kd_uint2 a = __builtin_bit_cast(kd_uint2, v.a[k]), b = __builtin_bit_cast(kd_uint2, v.b[k]);
        __builtin_amdgcn_raw_buffer_store_b128(kd_uint4{ a.x, a.y, b.x, b.y }, w, c.roff[k], 0, 0);
    }
    const unsigned ho = c.q == 0 ? c.hoff : KLARA_BUF_OOB;       // the hyper block is written once per chain
#pragma unroll
    for (int k = 0; k < 5; ++k)
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(kd_uint2, v.h[k]), w, ho + 8u * (unsigned)k, 0, 0);
}

// lt and/or gradient at th (oracle ko_hier_eval, operation for operation)
template <int RPL, int NT, bool WANT_LT, bool WANT_GRAD>
__device__ __forceinline__ double hier_eval(const HierLane<RPL, NT>& c, const HierVec<RPL>& th, HierVec<RPL>& g)
{
    const double ac = th.h[0], bc = th.h[1], sc = th.h[2], sa = th.h[3], sb = th.h[4];
    // exp(-2 s_k): lane q of the chain evaluates k = q (q < 3), the other lanes idle along; three broadcasts
    const int gb = c.lane - c.q;
    const double w = kd_exp(-2.0 * (c.q == 0 ? sc : (c.q == 1 ? sa : sb)));
    const double wc = lane_bcast(w, gb), wa = lane_bcast(w, gb + 1), wb = lane_bcast(w, gb + 2);
    double red[5] = { 0.0, 0.0, 0.0, 0.0, 0.0 };        // A1, B1, A2, B2, C2 lane partials over this lane's units
#pragma unroll
    for (int k = 0; k < RPL; ++k) {
        const double ai = th.a[k], bi = th.b[k];
        // a missing unit (the last lane of a chain when R % 4 != 0) is masked once, at its five per-unit quantities: its
        // sums and its gradient are then exactly 0, so its momentum and value stay 0 and it never enters a sum
        const double da = c.rv[k] ? ai - ac : 0.0, db = c.rv[k] ? bi - bc : 0.0;
        // residual sums r_j = y_j - a - b x_j from the unit's sufficient statistics (oracle ko_hier_eval):
        //   sum r = Sy - T a - b X1;  sum r x = Sxy - a X1 - b X2;  sum r^2 = Syy + a (T a - 2 Sy) + b (b X2 + 2 a X1 - 2 Sxy)
        // (missing unit: statistics = a = b = 0, so all three are 0)
        const double S1 = kd_fma(-bi, c.X1, kd_fma(-c.Td, ai, c.Sy[k]));
        const double Sx = kd_fma(-bi, c.X2, kd_fma(-ai, c.X1, c.Sxy[k]));
        const double u = kd_fma(c.Td, ai, c.m2Sy[k]);
        const double v = kd_fma(bi, c.X2, kd_fma(2.0 * ai, c.X1, c.m2Sxy[k]));
        const double S2 = kd_fma(bi, v, kd_fma(ai, u, c.Syy[k]));
        if (WANT_GRAD) { g.a[k] = kd_fma(wc, S1, -(wa * da)); g.b[k] = kd_fma(wc, Sx, -(wb * db)); }
        red[0] = red[0] + da;               red[1] = red[1] + db;
        red[2] = kd_fma(da, da, red[2]);    red[3] = kd_fma(db, db, red[3]);
        red[4] = red[4] + S2;
    }
    group_allreduce<5>(red, KLARA_HIERT_Q, c.lane);
    const double A1 = red[0], B1 = red[1], A2 = red[2], B2 = red[3], C2 = red[4];
    const double RT = (double)c.R * c.Td, Rd = (double)c.R;
    if (WANT_GRAD) {
        const double ta0 = 2.0 * c.a0, tb0 = 2.0 * c.b0;
        g.h[0] = kd_fma(wa, A1, -(c.p0 * ac));
        g.h[1] = kd_fma(wb, B1, -(c.p0 * bc));
        g.h[2] = kd_fma(tb0, wc, kd_fma(wc, C2, -RT) - ta0);
        g.h[3] = kd_fma(tb0, wa, kd_fma(wa, A2, -Rd) - ta0);
        g.h[4] = kd_fma(tb0, wb, kd_fma(wb, B2, -Rd) - ta0);
    }
    double lt = 0.0;
    if (WANT_LT) {
        const double l_c = (-RT * sc - 0.5 * (wc * C2)) + (-2.0 * c.a0 * sc - c.b0 * wc);
        const double l_a = (-Rd * sa - 0.5 * (wa * A2)) + (-2.0 * c.a0 * sa - c.b0 * wa);
        const double l_b = (-Rd * sb - 0.5 * (wb * B2)) + (-2.0 * c.a0 * sb - c.b0 * wb);
        lt = ((l_c + l_a) + l_b) - (0.5 * c.p0) * (ac * ac + bc * bc);
    }
    return lt;
}

// sum(p .* p) over the chain's D elements: lane partial over its units (a then b, ascending), lane 0 adds the five hyper
// terms, butterfly over the 8 lanes
template <int RPL, int NT>
__device__ __forceinline__ double hier_sumsq(const HierLane<RPL, NT>& c, const HierVec<RPL>& m)
{
    double s[1] = { 0.0 };
#pragma unroll
    for (int k = 0; k < RPL; ++k) { s[0] = s[0] + m.a[k] * m.a[k]; s[0] = s[0] + m.b[k] * m.b[k]; }
#pragma unroll
    for (int k = 0; k < 5; ++k) s[0] = s[0] + (c.q == 0 ? m.h[k] * m.h[k] : 0.0);
    group_allreduce<1>(s, KLARA_HIERT_Q, c.lane);
    return s[0];
}

// DA: DualAveragingMCTuner (iterate/HMC.jl:142-144, 225-249): per-chain step and per-chain trajectory length; the wavefront
// runs to the longest trajectory among its 8 chains and a finished chain's lanes keep their state (selects).
// sum over the chain's D elements of an elementwise term vector: same order as hier_sumsq
template <int RPL, int NT>
__device__ __forceinline__ double hier_sumvec(const HierLane<RPL, NT>& c, const HierVec<RPL>& tv)
{
    double s[1] = { 0.0 };
#pragma unroll
    for (int k = 0; k < RPL; ++k) { s[0] = s[0] + tv.a[k]; s[0] = s[0] + tv.b[k]; }
#pragma unroll
    for (int k = 0; k < 5; ++k) s[0] = s[0] + (c.q == 0 ? tv.h[k] : 0.0);
    group_allreduce<1>(s, KLARA_HIERT_Q, c.lane);
    return s[0];
}

// a per-element host vector of length D (proposal scales) as this lane sees it
template <int RPL, int NT>
__device__ __forceinline__ void hload_param(const HierLane<RPL, NT>& c, const gdouble* base, double dflt, HierVec<RPL>& v)
{
#pragma unroll
    for (int k = 0; k < RPL; ++k) {
        const int r = RPL * c.q + k;
        v.a[k] = (base != nullptr && c.rv[k]) ? base[2 * r] : dflt;
        v.b[k] = (base != nullptr && c.rv[k]) ? base[2 * r + 1] : dflt;
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) v.h[k] = base != nullptr ? base[2 * c.R + k] : dflt;
}

template <int SAMPLER, int RPL, int NT, bool MON, bool TUNE, bool DA = false>
__global__ __launch_bounds__(256, 2)
void k_hiert(const KParams* __restrict__ pp, const KLaunch kl)
{
    constexpr int Q = KLARA_HIERT_Q, CPW = 64 / Q;
    static_assert(!DA || (TUNE && SAMPLER == KLARA_SAMPLER_HMC), "dual averaging is a tuned HMC instantiation");
    static_assert(SAMPLER == KLARA_SAMPLER_MH || SAMPLER == KLARA_SAMPLER_MALA || SAMPLER == KLARA_SAMPLER_HMC, "sampler");
    constexpr bool NEEDG = SAMPLER != KLARA_SAMPLER_MH;
    constexpr bool PLAIN = !TUNE;
    const KParams& p = *pp;
    kd_tables_to_lds();
    const HierLane<RPL, NT> cx = make_hlane<RPL, NT>(p);
    const int D = p.D, R = p.hR;
    guchar* const accept_out = p.accept != nullptr ? p.accept + kl.t0 * (unsigned long long)p.nchains : nullptr;

    const long long grp = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));   // wave-uniform: scalar windows
    if (grp * CPW >= p.nchains) return;
    const long long first_chain = grp * CPW;
    const long long left = p.nchains - first_chain;
    const int here = left < CPW ? (int)left : CPW;
    const bool chain_ok = cx.cw < here;
    const long long chain = first_chain + cx.cw;
    const unsigned long long gchain = (unsigned long long)(p.chain_offset + chain);
    const __amdgpu_buffer_rsrc_t wx = group_window(p.X, first_chain, here, D);
    const __amdgpu_buffer_rsrc_t wg = group_window(p.GR, first_chain, here, D);

    HierVec<RPL> x, g, sig;
    hload<RPL, NT>(cx, wx, x);
    if (NEEDG) hload<RPL, NT>(cx, wg, g);
    if (SAMPLER == KLARA_SAMPLER_MH) hload_param<RPL, NT>(cx, p.vecparam, 1.0, sig);
    const long long c0 = chain_ok ? chain : 0;
    double lt = p.LT[c0];
    unsigned long long nacc = 0;
    // Running sums (sojourn form, KParams::held) live in LDS for the launch, one private column per lane — [wavefront][slot][lane],
    // 2 * (2 RPL + 5) slots, conflict-free 8-byte accesses — instead of 4 RPL + 10 registers per lane that the 256-register budget
    // of two wavefronts per SIMD does not have (they were spilled: 272 B of scratch).  They are touched when a chain leaves a state.
    constexpr int NSLOT = 2 * RPL + 5;
    __shared__ double lds_sums[MON ? 4 * 2 * NSLOT * 64 : 1];
    double* const my_sums = &lds_sums[MON ? ((threadIdx.x >> 6) * 2 * NSLOT) * 64 + (threadIdx.x & 63) : 0];
    const bool do_sum = MON && p.sum != nullptr;
    long long held = 0;
    if (do_sum) {
        HierVec<RPL> sm, sq;
        hload<RPL, NT>(cx, group_window(p.sum, first_chain, here, D), sm);
        hload<RPL, NT>(cx, group_window(p.sumsq, first_chain, here, D), sq);
#pragma unroll
        for (int k = 0; k < RPL; ++k) {
            my_sums[(2 * k) * 64] = sm.a[k]; my_sums[(2 * k + 1) * 64] = sm.b[k];
            my_sums[(NSLOT + 2 * k) * 64] = sq.a[k]; my_sums[(NSLOT + 2 * k + 1) * 64] = sq.b[k];
        }
#pragma unroll
        for (int k = 0; k < 5; ++k) { my_sums[(2 * RPL + k) * 64] = sm.h[k]; my_sums[(NSLOT + 2 * RPL + k) * 64] = sq.h[k]; }
        held = p.held[chain_ok ? chain : 0];
    }
    int sphase = kl.save_phase0;
    long long scol = kl.save_col0;
    const bool per_chain_tune = KCNT && !KPOOLED;
    TuneRegs tn;
    if (per_chain_tune) tn = { p.tune_step[c0], p.tune_accepted[c0], p.tune_proposed[c0], p.tune_totproposed[c0], 0, 0.0, 0.0 };
    else if (KPOOLED) tn = { p.tune_step[0], p.tune_accepted[0], 0, 0, 0, 0.0, 0.0 };
    else tn = { DA ? p.tune_step[c0] : p.step0, 0, 0, 0, 0, 0.0, 0.0 };
    if (DA) { tn.epsbar = p.da_epsbar[c0]; tn.hbar = p.da_hbar[c0]; }
    const long long acc0 = tn.accepted;
    tn.phase = per_chain_tune ? (int)(tn.proposed % p.period) : 0;

    for (int s = 0; s < kl.nsteps; ++s) {
        const unsigned long long t = kl.t0 + (unsigned long long)s;
        if (KCNT) tune_count_proposal(p, tn);
        // momentum ~ N(0, I) (iterate/HMC.jl:135): unit i = elements 2i, 2i+1 = pair index i; the hyper block (elements 2R..2R+4) is pair
        // indices R, R+1, R+2 (64 bits of a Philox block each: kd_normal_pair_at); the accept uniform is pair index ceil(D/2) = R + 3
        HierVec<RPL> mom;
#pragma unroll
        for (int k = 0; k < RPL; ++k) {
            { double u1_, lg_; kd_normal_pair_at(p.seed, gchain, t, (uint32_t)(RPL * cx.q + k), (uint32_t)(R + 3), &mom.a[k], &mom.b[k], &u1_, &lg_); }
            if (!cx.rv[k]) { mom.a[k] = 0.0; mom.b[k] = 0.0; }
            __builtin_amdgcn_sched_barrier(0);
        }
        // The hyper block's three blocks and the block of the accept draw (slot R + 3 = ceil(D/2)) are the same for the 8 lanes of a chain:
        // lane q evaluates ONE of them — slot R + (q & 3) — and the four lanes of a quad exchange the results (2 v_mov_dpp per double)
        // instead of every lane evaluating all four: 102 + 14 vector instructions per transition instead of 3 x 102 + 45.
        double acc_u, acc_logu;
        {
            double z0, z1, u1, lg1;
            kd_normal_pair_at(p.seed, gchain, t, (uint32_t)(R + (cx.q & 3)), (uint32_t)(R + 3), &z0, &z1, &u1, &lg1);
            mom.h[0] = quad_bcast(z0, 0); mom.h[1] = quad_bcast(z1, 0);
            mom.h[2] = quad_bcast(z0, 1); mom.h[3] = quad_bcast(z1, 1);
            mom.h[4] = quad_bcast(z0, 2);
            acc_u = quad_bcast(u1, 3); acc_logu = quad_bcast(lg1, 3);
        }
        HierVec<RPL> xp, gp;
        double ltp, a = 0.0;
        bool acc;
        if (SAMPLER == KLARA_SAMPLER_HMC) {
            const double H0 = lt - 0.5 * hier_sumsq<RPL, NT>(cx, mom);                     // :137
            xp = x; gp = g;                                                               // :139-140
            const double eps = tn.step, halfe = 0.5 * eps;
            const int nl = DA ? (chain_ok ? da_nleaps(p, eps) : 1) : p.nleaps;            // iterate/HMC.jl:142-144 (padding lanes: 1)
            // leapfrog! L times (:146-155, samplers.jl:122-134) in its merged form — DESIGN.md section 2, deliberate deviation (7),
            // the oracle takes the same steps for this layout: the closing half-kick of step l and the opening half-kick of step
            // l + 1 use the same gradient and are ONE update p += eps g, and every update is one fma:
            //   p = fma(eps/2, g, p);  L x { x = fma(eps, p, x);  g = grad(x);  p = fma(l < L-1 ? eps : eps/2, g, p) }
            // 26 instead of 78 vector instructions per step for the kicks and the drift of a lane's 13 values.
    #pragma unroll
            for (int k = 0; k < RPL; ++k) { mom.a[k] = kd_fma(halfe, gp.a[k], mom.a[k]); mom.b[k] = kd_fma(halfe, gp.b[k], mom.b[k]); }
    #pragma unroll
            for (int k = 0; k < 5; ++k) mom.h[k] = kd_fma(halfe, gp.h[k], mom.h[k]);
            const int nlmax = DA ? wave_max_int(nl) : nl;
            for (int l = 0; l < nlmax; ++l) {
                const bool go = !DA || l < nl;                                            // (a finished chain keeps its state)
    #pragma unroll
                for (int k = 0; k < RPL; ++k) {
                    const double xa = kd_fma(eps, mom.a[k], xp.a[k]), xb = kd_fma(eps, mom.b[k], xp.b[k]);
                    xp.a[k] = go ? xa : xp.a[k]; xp.b[k] = go ? xb : xp.b[k];
                }
    #pragma unroll
                for (int k = 0; k < 5; ++k) { const double xh = kd_fma(eps, mom.h[k], xp.h[k]); xp.h[k] = go ? xh : xp.h[k]; }
                HierVec<RPL> gn;
                (void)hier_eval<RPL, NT, false, true>(cx, xp, gn);
                const double kf = l + 1 < nl ? eps : halfe;                               // the chain's last step closes with a half-kick
    #pragma unroll
                for (int k = 0; k < RPL; ++k) {
                    gp.a[k] = go ? gn.a[k] : gp.a[k]; gp.b[k] = go ? gn.b[k] : gp.b[k];
                    const double ma = kd_fma(kf, gp.a[k], mom.a[k]), mb = kd_fma(kf, gp.b[k], mom.b[k]);
                    mom.a[k] = go ? ma : mom.a[k]; mom.b[k] = go ? mb : mom.b[k];
                }
    #pragma unroll
                for (int k = 0; k < 5; ++k) {
                    gp.h[k] = go ? gn.h[k] : gp.h[k];
                    const double mh = kd_fma(kf, gp.h[k], mom.h[k]);
                    mom.h[k] = go ? mh : mom.h[k];
                }
            }
            HierVec<RPL> gdummy;
            ltp = hier_eval<RPL, NT, true, false>(cx, xp, gdummy);                        // :157
            const double H1 = ltp - 0.5 * hier_sumsq<RPL, NT>(cx, mom);                   // :159
            const double ratio = H1 - H0;                                                 // :161
            const double ex = kd_exp(ratio);
            a = 1.0 < ex ? 1.0 : ex;                                                      // :163
            acc = acc_u < a;                                                              // :165
        } else if (SAMPLER == KLARA_SAMPLER_MALA) {                                   // iterate/MALA.jl:78-128 (mom holds z)
            const double h_ = tn.step, halfh = 0.5 * h_, sqh = KCNT ? __builtin_sqrt(h_) : p.sqrt_step0;
            const double half_inv_h = 0.5 * (KCNT ? 1.0 / h_ : p.inv_step0);
#pragma unroll
            for (int k = 0; k < RPL; ++k) {
                xp.a[k] = (x.a[k] + halfh * g.a[k]) + sqh * mom.a[k];                          // :83-84
                xp.b[k] = (x.b[k] + halfh * g.b[k]) + sqh * mom.b[k];
            }
#pragma unroll
            for (int k = 0; k < 5; ++k) xp.h[k] = (x.h[k] + halfh * g.h[k]) + sqh * mom.h[k];
            ltp = hier_eval<RPL, NT, true, true>(cx, xp, gp);                                  // :86
            // the two proposal-density terms, summed in hier_sumvec's order (a_k, b_k ascending, then the hyper block on lane 0); the
            // drift mean x + h/2 g is formed again instead of being kept across the evaluation (same operations, same bits)
            const auto qterm = [&](double from_x, double from_g, double to) {
                const double d = (from_x + halfh * from_g) - to;
                return (d * d) * half_inv_h;
            };
            double s1[1] = { 0.0 }, s2[1] = { 0.0 };
#pragma unroll
            for (int k = 0; k < RPL; ++k) {
                s1[0] = s1[0] + qterm(x.a[k], g.a[k], xp.a[k]); s1[0] = s1[0] + qterm(x.b[k], g.b[k], xp.b[k]);     // :90
                s2[0] = s2[0] + qterm(xp.a[k], gp.a[k], x.a[k]); s2[0] = s2[0] + qterm(xp.b[k], gp.b[k], x.b[k]);   // :91-92
            }
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const double q1 = qterm(x.h[k], g.h[k], xp.h[k]), q2 = qterm(xp.h[k], gp.h[k], x.h[k]);
                s1[0] = s1[0] + (cx.q == 0 ? q1 : 0.0); s2[0] = s2[0] + (cx.q == 0 ? q2 : 0.0);
            }
            group_allreduce<1>(s1, KLARA_HIERT_Q, cx.lane);
            group_allreduce<1>(s2, KLARA_HIERT_Q, cx.lane);
            double ratio = ltp - lt;                                                           // :88
            ratio += s1[0];                                                                    // :90
            ratio -= s2[0];                                                                    // :92
            acc = ratio > 0.0;                                                                 // :94
            acc = acc || ratio > acc_logu;
        } else {                                                                       // iterate/MH.jl:72-124 (mom holds z)
#pragma unroll
            for (int k = 0; k < RPL; ++k) { xp.a[k] = x.a[k] + sig.a[k] * mom.a[k]; xp.b[k] = x.b[k] + sig.b[k] * mom.b[k]; }   // :79
#pragma unroll
            for (int k = 0; k < 5; ++k) xp.h[k] = x.h[k] + sig.h[k] * mom.h[k];
            ltp = hier_eval<RPL, NT, true, false>(cx, xp, gp);                                 // :81
            const double ratio = ltp - lt;                                                     // :83
            acc = ratio > 0.0;                                                                 // :97
            acc = acc || ratio > acc_logu;
        }
        if (do_sum && __any(acc && held > 0)) {                                       // leaving a state after `held` saved steps
            const bool fold = acc && held > 0;
            const double hf = fold ? (double)held : 0.0;
            if (fold) {
#pragma unroll
                for (int k = 0; k < RPL; ++k) {
                    my_sums[(2 * k) * 64] = my_sums[(2 * k) * 64] + hf * x.a[k];
                    my_sums[(2 * k + 1) * 64] = my_sums[(2 * k + 1) * 64] + hf * x.b[k];
                    my_sums[(NSLOT + 2 * k) * 64] = my_sums[(NSLOT + 2 * k) * 64] + hf * (x.a[k] * x.a[k]);
                    my_sums[(NSLOT + 2 * k + 1) * 64] = my_sums[(NSLOT + 2 * k + 1) * 64] + hf * (x.b[k] * x.b[k]);
                }
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    my_sums[(2 * RPL + k) * 64] = my_sums[(2 * RPL + k) * 64] + hf * x.h[k];
                    my_sums[(NSLOT + 2 * RPL + k) * 64] = my_sums[(NSLOT + 2 * RPL + k) * 64] + hf * (x.h[k] * x.h[k]);
                }
                held = 0;
            }
        }
        if (acc) { x = xp; if (NEEDG) g = gp; lt = ltp; }                             // commit (HMC.jl:166-176, MALA.jl:95-105, MH.jl:98-100)
        nacc += acc ? 1ull : 0ull;
        if (accept_out != nullptr && chain_ok && cx.q == 0) accept_out[(long long)s * p.nchains + chain] = acc ? 1 : 0;
        if (KCNT && acc) tn.accepted += 1;
        if (DA) da_update(p, tn, (long long)t + 1, a);                                // iterate/HMC.jl:225-249
        if (per_chain_tune && !DA) tuning_block(p, tn);                               // iterate/HMC.jl:203-224, MALA.jl:130-152
        else if (DA && per_chain_tune && tn.phase == 0 && (long long)t + 1 <= p.da_nadapt) {   // verbose report block, :229-243
            tn.totproposed += tn.proposed; tn.accepted = 0; tn.proposed = 0;
        }
        // save rule: BasicMCJob.jl:226-231 with postrange = (burnin+1):thinning:nsteps (BasicMCRange.jl:36)
        const long long i1 = (long long)t + 1;
        if (MON && i1 > p.burnin && i1 <= p.nsteps_total) {
            if (sphase == 0) {
                if (do_sum) held += 1;
                if (scol < p.hist_cols) {
                    const long long col0 = scol * p.nchains + first_chain;
                    if (p.hist != nullptr) hstore<RPL, NT>(cx, group_window(p.hist, col0, here, D), x);
                    if (NEEDG && p.hist_g != nullptr) hstore<RPL, NT>(cx, group_window(p.hist_g, col0, here, D), g);
                    if (p.hist_lt != nullptr && chain_ok && cx.q == 0) p.hist_lt[scol * p.nchains + chain] = lt;
                }
                ++scol;
            }
            sphase = (sphase + 1 == (int)p.thinning) ? 0 : sphase + 1;
        }
    }
    if (do_sum) {
        HierVec<RPL> sm, sq;
#pragma unroll
        for (int k = 0; k < RPL; ++k) {
            sm.a[k] = my_sums[(2 * k) * 64]; sm.b[k] = my_sums[(2 * k + 1) * 64];
            sq.a[k] = my_sums[(NSLOT + 2 * k) * 64]; sq.b[k] = my_sums[(NSLOT + 2 * k + 1) * 64];
        }
#pragma unroll
        for (int k = 0; k < 5; ++k) { sm.h[k] = my_sums[(2 * RPL + k) * 64]; sq.h[k] = my_sums[(NSLOT + 2 * RPL + k) * 64]; }
        hstore<RPL, NT>(cx, group_window(p.sum, first_chain, here, D), sm);
        hstore<RPL, NT>(cx, group_window(p.sumsq, first_chain, here, D), sq);
        if (chain_ok && cx.q == 0) p.held[chain] = held;
    }
    if (nacc != 0) {
        hstore<RPL, NT>(cx, wx, x);
        if (NEEDG) hstore<RPL, NT>(cx, wg, g);
        if (chain_ok && cx.q == 0) { p.LT[chain] = lt; p.naccept[chain] += nacc; }
    }
    if (TUNE && chain_ok && cx.q == 0) {
        if (DA) { p.tune_step[chain] = tn.step; p.da_epsbar[chain] = tn.epsbar; p.da_hbar[chain] = tn.hbar; }
        if (per_chain_tune) {
            p.tune_step[chain] = tn.step; p.tune_accepted[chain] = tn.accepted;
            p.tune_proposed[chain] = tn.proposed; p.tune_totproposed[chain] = tn.totproposed;
        } else if (KPOOLED && KCNT) {
            atomicAdd((unsigned long long*)p.pooled_accepted, (unsigned long long)(tn.accepted - acc0));
        }
    }
}

// initialize!(pstate, parameter, sampler) for layout kind 4: lt and gradient at X, finiteness check (HMC.jl:106-120)
template <int RPL, int NT>
__global__ __launch_bounds__(256) void k_hiert_init(const KParams p, int needgrad)
{
    constexpr int CPW = 64 / KLARA_HIERT_Q;
    kd_tables_to_lds();
    const HierLane<RPL, NT> cx = make_hlane<RPL, NT>(p);
    const long long grp = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));   // wave-uniform: scalar windows
    const long long first_chain = grp * CPW;
    const long long left = p.nchains - first_chain;
    const int here = left < CPW ? (left > 0 ? (int)left : 0) : CPW;
    const bool chain_ok = cx.cw < here;
    const long long chain = first_chain + cx.cw;
    HierVec<RPL> x, g;
    hload<RPL, NT>(cx, group_window(p.X, first_chain, here, p.D), x);
    const double lt = hier_eval<RPL, NT, true, true>(cx, x, g);
    bool bad = !kfinite(lt);
    if (needgrad) {
        hstore<RPL, NT>(cx, group_window(p.GR, first_chain, here, p.D), g);
#pragma unroll
        for (int k = 0; k < RPL; ++k) bad = bad || (cx.rv[k] && (!kfinite(g.a[k]) || !kfinite(g.b[k])));
#pragma unroll
        for (int k = 0; k < 5; ++k) bad = bad || !kfinite(g.h[k]);
    }
    if (chain_ok && cx.q == 0) p.LT[chain] = lt;
    if (chain_ok && bad) klara_raise(p.error_flag, KLARA_ERR_NONFINITE_INIT);
}
