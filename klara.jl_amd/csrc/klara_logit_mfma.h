// klara_logit_mfma.h — the logistic regression beyond 16 parameters on the FP64 matrix cores (layout kind 5; round 6).
//
// doc/examples/swiss/MALA/analytical.jl:11-18:  ll = (X p).y - sum log(1 + exp(X p)),  lp = -(p.p / lambda + D log(2 pi lambda)) / 2,
// grad = X' (y - 1 ./ (1 + exp(-X p))) - p / lambda.  With D parameters and n data rows both products are DENSE CONTRACTIONS over many
// chains at once — X (n x D) times the parameters of 16 chains, X' (D x n) times their residuals — i.e. what north_star reserves the matrix
// cores for.  Up to 16 parameters the row-split kernels (klara_kernels.h LogisticTarget: every lane holds the whole vector) are faster; beyond
// them rounds 1-5 ran the target as a closure, one chain per lane with scalar loops over rows and columns (1.6e8 transitions/s at D = 20,
// 4.7e7 at D = 64 against 9e8 at D = 16: profiles/r6_logit_mfma.txt).
//
// Layout: klara_dense.h's.  A wavefront carries 16 chains; lane (q = l >> 4, c = l & 15) holds the elements {4 e + q} of chain c, NE = 8,
// 16, 24, 32 per lane (D <= 32, 64, 96, 128) and 40, 48, 56, 64 (D <= 160 .. 256: one wavefront per SIMD).  One evaluation runs over the data rows in BLOCKS of RBT tiles of 16 rows:
//   pass 1   Z = X P        v_mfma_f64_16x16x4:  A = X[16 T + (l & 15)][4 kk + (l >> 4)],  B = the lane's own element kk,  kk = 0 .. NE-1
//            -> accumulator tile T, register j of lane (q, c) = (X p)[row 16 T + 4 j + q] of chain c
//   rows     the lane's 4 RBT row values: softplus / logistic from ONE exponential (detmath.h kd_softplus_logistic_rows, the row arithmetic of
//            every logistic kernel and of the oracle), the residual y - 1/(1 + exp(-Xp)) written back into the same registers
//   pass 2   G += X' R       A = X[16 T + 4 j + (l >> 4)][16 t + (l & 15)],  B = the residual register j of tile T — the accumulator layout of
//            pass 1 IS the B-operand layout of pass 2 (klara_dense.h's trick: no transpose, no LDS round trip) -> gradient tile t, register r =
//            element 4 t + r of the lane, accumulated over all blocks.
// The A fragments of both passes are STREAMED from memory in the order of consumption through a register ring (klara_dense_big.h): every
// wavefront of every compute unit reads the same stream, so it lives in the L2s / L1s; n is not limited by the LDS.  One wavefront per SIMD, the
// momentum / proposal normals / current value in the lane's LDS column, as in klara_dense_big.h.
//
// Summation orders (the oracle's ko_logit_eval, layout kind 5): X p and X' r are the fma chains of the MFMA (k ascending from zero: columns,
// respectively data rows) — the chains of the closure form, bit for bit; the row sums (X p).y and sum log(1 + exp) are lane partials over the
// lane's rows ascending, then (q0 + q1) + (q2 + q3); p.p likewise over the lane's elements.  Padding rows (>= n) and columns (>= D) are zeros in
// the stream: they add exact zeros to every chain; a padding row's softplus term is masked.
// MH, MALA, HMC (Vanilla / AcceptanceRate per chain or pooled / dual averaging) and the slice sampler (a probe = pass 1 + the rows; the chains of a tile out
// of lockstep), every monitor of klara_dense_big.h.  The likelihood / prior history keeps the closure form.
#pragma once
#include "klara_dense_big.h"

#ifndef KLARA_LOGITM_RBT
#define KLARA_LOGITM_RBT 2            // row tiles (16 data rows each) per block: 8 row values per lane in flight
#endif
#ifndef KLARA_LOGITM_RING
#define KLARA_LOGITM_RING 16          // fragments in flight per lane (16 MFMAs = 1,024 cycles ahead of their use: an L2 round trip) ...
#endif
#ifndef KLARA_LOGITM_RING_BIG
#define KLARA_LOGITM_RING_BIG 8       // ... and at NE = 24 / 32, where 16 more registers are 16 more values in scratch
#endif

// One evaluation at x (the lane's elements): ga = X' (y - logistic(X x)) as MFMA tiles (WANT_G), and the lane's partial row sums.
// F: the fragment stream; block b holds 2 S1 fragments of 64 doubles: S1 = RBT NE of pass 1 (step kk RBT + tt), then S1 of pass 2
// (step (4 tt + j) MT + t); ypad: the responses, zero-padded to the blocks' rows.
// The row arithmetic of R values in STAGES (klara_kernels.h LogisticTarget::eval rows_of: reduce -> table gather -> polynomial -> combine -> bin -> table
// gather -> division -> finish, every stage for all R before the next): kd_softplus_logistic_rows (detmath.h) operation for operation, with the R table
// gathers of a stage in flight together instead of one exposed LDS round trip after the other.  sp = log(1 + exp(v)), lg = 1 / (1 + exp(-v)).
template <int R>
__device__ __forceinline__ void logitm_rows(const double* sL12, const double (&v)[R], double (&sp)[R], double (&lg)[R])
{
    double rr[R], th[R], tl[R], t[R], onept[R], invc[R], logc[R];
    int kk[R]; uint32_t li[R];
#pragma unroll
    for (int j = 0; j < R; ++j) kd_exp_neg_reduce(__builtin_fabs(v[j]), &kk[j], &rr[j]);                      // t = exp(-|Xp|): one exponential for both functions
    KLARA_SCHED_STAGE();
#pragma unroll
    for (int j = 0; j < R; ++j) { const int idx = kk[j] & 127; th[j] = KD_EXPTAB(2 * idx); tl[j] = KD_EXPTAB(2 * idx + 1); }
    KLARA_SCHED_STAGE();
#pragma unroll
    for (int j = 0; j < R; ++j) { KLARA_PIN(rr[j]); rr[j] = kd_exp_neg_poly(rr[j]); }                         // (needs no table value: issued under the gathers)
    KLARA_SCHED_STAGE();
#pragma unroll
    for (int j = 0; j < R; ++j) { t[j] = kd_exp_neg_combine(kk[j], rr[j], th[j], tl[j]); onept[j] = 1.0 + t[j]; li[j] = kd_log12_bin(onept[j]); }
    KLARA_SCHED_STAGE();
#pragma unroll
    for (int j = 0; j < R; ++j) { invc[j] = sL12[2 * li[j]]; logc[j] = sL12[2 * li[j] + 1]; }
    KLARA_SCHED_STAGE();
#pragma unroll
    for (int j = 0; j < R; ++j) lg[j] = kd_div_unit_range(v[j] >= 0.0 ? 1.0 : t[j], onept[j]);                // 1/(1+exp(-Xp)) (no table value either)
    KLARA_SCHED_STAGE();
#pragma unroll
    for (int j = 0; j < R; ++j) sp[j] = (v[j] > 0.0 ? v[j] : 0.0) + kd_log12_finish(onept[j], invc[j], logc[j]);   // log(1+exp(Xp))
    KLARA_SCHED_STAGE();
}

template <int NE, bool WANT_G>
__device__ __forceinline__ void logitm_eval(const double* __restrict__ F, const gdouble* __restrict__ ypad, const double* sL12, int ndata, int nblocks, int lane,
                                            const double (&x)[NE], kd_double4 (&ga)[NE / 4], double& dotxy, double& slog)
{
    constexpr int MT = NE / 4, RBT = KLARA_LOGITM_RBT, S1 = RBT * NE, SB = WANT_G ? 2 * S1 : S1, RING = NE <= 16 ? KLARA_LOGITM_RING : KLARA_LOGITM_RING_BIG;
    static_assert(NE % 4 == 0 && S1 % RING == 0 && S1 >= RING, "whole tiles; a block is a whole number of rings");
    const int q = lane >> 4;
    if (WANT_G) {
#pragma unroll
        for (int t = 0; t < MT; ++t) ga[t] = (kd_double4){ 0.0, 0.0, 0.0, 0.0 };
    }
    double sxy = 0.0, slg = 0.0;
    // The stream and the responses through buffer resources: the lane's byte offset is ONE register for the whole evaluation (lane x 8, q x 8), the
    // position in the stream is a scalar offset — no vector address arithmetic per fragment.  (Behind the empty asm the offset is "new" in every call: the
    // loads stay where they are written — klara_dense_big.h dense_stream.)
    const __amdgpu_buffer_rsrc_t rF = __builtin_amdgcn_make_buffer_rsrc((void*)F, 0, __builtin_amdgcn_readfirstlane(nblocks) * (2 * S1 * 512), 0x00020000);
    const __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc((void*)ypad, 0, __builtin_amdgcn_readfirstlane(nblocks) * (RBT * 16 * 8), 0x00020000);
    unsigned voff = (unsigned)lane * 8u, yoff = (unsigned)q * 8u;
    __asm__ volatile("" : "+v"(voff), "+v"(yoff));
    const auto frag = [&](int sbytes) { return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rF, voff, sbytes, 0)); };
    double ring[RING];
#pragma unroll
    for (int i = 0; i < RING; ++i) ring[i] = frag(i * 512);
    for (int b = 0; b < nblocks; ++b) {
        const int blk = __builtin_amdgcn_readfirstlane(b) * (2 * S1 * 512);          // the block's first byte in the stream (scalar)
        const bool more = b + 1 < nblocks;
        // the responses of the lane's rows of this block: row 16 (b RBT + tt) + 4 j + q  (in flight under pass 1)
        double yv[RBT * 4];
        const int yblk = __builtin_amdgcn_readfirstlane(b) * (RBT * 16 * 8);
#pragma unroll
        for (int i = 0; i < RBT * 4; ++i) yv[i] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rY, yoff, yblk + (16 * (i >> 2) + 4 * (i & 3)) * 8, 0));
        kd_double4 z[RBT];
#pragma unroll
        for (int tt = 0; tt < RBT; ++tt) z[tt] = (kd_double4){ 0.0, 0.0, 0.0, 0.0 };
        // -- pass 1: Z = X P over the block's row tiles
#pragma unroll
        for (int s = 0; s < S1; ++s) {
            const int kk = s / RBT, tt = s % RBT;
            const double a = ring[s % RING];
            if (s + RING < SB) ring[s % RING] = frag(blk + (s + RING) * 512);
            else if (more) ring[s % RING] = frag(blk + (2 * S1 + s + RING - SB) * 512);        // (lt only: the next block's pass 1)
            z[tt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, x[kk], z[tt], 0, 0, 0);
            if (tt == RBT - 1) __builtin_amdgcn_sched_barrier(0);
        }
        // -- the lane's rows of the block, ascending: tile tt, register j = row 16 (b RBT + tt) + 4 j + q
        const int row0 = b * (RBT * 16) + q;
        // (R row values per staged batch: 8 where the registers are there, 4 at NE = 24 / 32 — a batch keeps ~20 registers per value alive)
#ifndef KLARA_LOGITM_ROWS_SMALL
#define KLARA_LOGITM_ROWS_SMALL (KLARA_LOGITM_RBT * 4)
#endif
        constexpr int R = NE <= 16 ? KLARA_LOGITM_ROWS_SMALL : 4;
#pragma unroll
        for (int i0 = 0; i0 < RBT * 4; i0 += R) {
            double zv[R], sp[R], lg[R];
#pragma unroll
            for (int i = 0; i < R; ++i) zv[i] = z[(i0 + i) >> 2][(i0 + i) & 3];
            logitm_rows<R>(sL12, zv, sp, lg);
#pragma unroll
            for (int i = 0; i < R; ++i) {                              // the accumulations last, row by row in ascending order
                const int ii = i0 + i;
                const bool valid = row0 + 16 * (ii >> 2) + 4 * (ii & 3) < ndata;
                sxy = sxy + zv[i] * yv[ii];                            // dot(Xp, y)   (a padding row: +0 * 0)
                slg = slg + (valid ? sp[i] : 0.0);                     // sum(log(1 + exp(Xp)))
                if (WANT_G) z[ii >> 2][ii & 3] = valid ? yv[ii] - lg[i] : 0.0;   // y - 1/(1 + exp(-Xp)): pass 2's B operand, in place
            }
        }
        // -- pass 2: G += X' R
        if constexpr (WANT_G) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s2 = 0; s2 < S1; ++s2) {
                const int s = S1 + s2, tt = s2 / (4 * MT), j = (s2 / MT) & 3, t = s2 % MT;
                const double a = ring[s % RING];
                if (s + RING < SB) ring[s % RING] = frag(blk + (s + RING) * 512);
                else if (more) ring[s % RING] = frag(blk + (2 * S1 + s + RING - SB) * 512);
                ga[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, (double)z[tt][j], ga[t], 0, 0, 0);
                if (t == MT - 1) __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    dotxy = sxy; slog = slg;
}

// log-target and gradient at x from one evaluation: lt = (Xp.y - sum softplus) + -(p.p / lambda + lpconst) / 2, g = X' res - p / lambda
template <int NE, bool WANT_G>
__device__ __forceinline__ double logitm_target(const KParams& p, const double* __restrict__ F, const gdouble* __restrict__ ypad, const double* sL12, int nblocks, int lane,
                                                const double (&x)[NE], double (&g)[NE])
{
    kd_double4 ga[NE / 4];
    double red[3];
    logitm_eval<NE, WANT_G>(F, ypad, sL12, p.ndata, nblocks, lane, x, ga, red[0], red[1]);
    double pp = 0.0;
#pragma unroll
    for (int e = 0; e < NE; ++e) pp = pp + x[e] * x[e];
    red[2] = pp;
    mreduce<3>(red, lane);
    if (WANT_G) {
        const double lam = p.lambda;
#pragma unroll
        for (int e = 0; e < NE; ++e) g[e] = (double)ga[e >> 2][e & 3] - x[e] / lam;        // -p/v[1]
    }
    const double ll = red[0] - red[1];
    const double lp = -0.5 * (red[2] / p.lambda + p.lpconst);                               // plogprior
    return ll + lp;
}

// wavefronts per SIMD the register allocator leaves room for: the rows' arithmetic (~60 dependent-ish instructions per row value, two table gathers) is
// what a wavefront alone on its SIMD cannot keep issuing; two per SIMD measured fastest at every NE (profiles/r6_logit_mfma.txt; KLARA_LOGITM_WAVES_<NE>: A/B builds)
#ifndef KLARA_LOGITM_WAVES_8
#define KLARA_LOGITM_WAVES_8 2
#endif
#ifndef KLARA_LOGITM_WAVES_16
#define KLARA_LOGITM_WAVES_16 2
#endif
#ifndef KLARA_LOGITM_WAVES_24
#define KLARA_LOGITM_WAVES_24 2
#endif
#ifndef KLARA_LOGITM_WAVES_32
#define KLARA_LOGITM_WAVES_32 2
#endif
// (NE = 40 .. 64 — 129 .. 256 parameters — run one wavefront per SIMD like the streamed dense kernels: value and gradient alone are 256 registers at NE = 64)
template <int NE> __host__ __device__ constexpr int logitm_waves() { return NE <= 8 ? KLARA_LOGITM_WAVES_8 : NE <= 16 ? KLARA_LOGITM_WAVES_16 : NE <= 24 ? KLARA_LOGITM_WAVES_24 : NE <= 32 ? KLARA_LOGITM_WAVES_32 : 1; }

template <int SAMPLER, int NE, bool DA = false>
__global__ __launch_bounds__(256, logitm_waves<NE>())
void k_logit_mfma(const KParams* __restrict__ pp, const KLaunch kl, const double* __restrict__ F, const double* __restrict__ ypad_, int nblocks)
{
    static_assert(SAMPLER == KLARA_SAMPLER_HMC || SAMPLER == KLARA_SAMPLER_MALA || SAMPLER == KLARA_SAMPLER_MH || SAMPLER == KLARA_SAMPLER_SLICE, "HMC, MALA, MH, slice");
    constexpr bool SLICE = SAMPLER == KLARA_SAMPLER_SLICE;
    constexpr bool XMEM = SAMPLER == KLARA_SAMPLER_HMC || SLICE;      // the value a chain is leaving / goes back to is read from X (MALA / MH: from the lane's LDS column)
    static_assert(!DA || SAMPLER == KLARA_SAMPLER_HMC, "dual averaging is wired into HMC only (HMC.jl:124-133)");
    constexpr bool NEEDG = SAMPLER == KLARA_SAMPLER_MALA || SAMPLER == KLARA_SAMPLER_HMC;      // MH and the slice sampler carry no gradient (GR is not written)
    constexpr bool da = DA;
    const KParams& p = *pp;
    const gdouble* const ypad = (const gdouble*)ypad_;
    guchar* const accept_out = p.accept != nullptr ? p.accept + kl.t0 * (unsigned long long)p.nchains : nullptr;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // MH's proposal scales, sigma[4 e + q] at [4 e + q], 0 past D
    constexpr bool SIGLDS = SAMPLER == KLARA_SAMPLER_MH;
    double* const ldsL12 = reinterpret_cast<double*>(smem);          // the rows' log(1 + t) table (kd_log12): gathered from LDS like the other tables
    for (int i = threadIdx.x; i < 256; i += blockDim.x) ldsL12[i] = kd_l12tab_dev[i];
    double* const ldsSig = ldsL12 + 256;
    if (SIGLDS) { for (int i = threadIdx.x; i < 4 * NE; i += blockDim.x) ldsSig[i] = (p.vecparam != nullptr && i < p.D) ? p.vecparam[i] : 0.0; }
    kd_tables_to_lds();          // (also the barrier for sigma)
    const MfmaCtx<NE> cx = make_mctx<NE>(p);
    // this lane's LDS column: momentum (HMC), the proposal's normals then the current value (MALA), the current value (MH)
    double* const momw = ldsSig + (SIGLDS ? 4 * NE : 0) + (size_t)(threadIdx.x >> 6) * NE * 64 + cx.lane;
    const unsigned long long gchain = (unsigned long long)(p.chain_offset + cx.chain);
    const long long tix = p.pooled ? 0 : (cx.chain_ok ? cx.chain : 0);
    TuneRegs tn = { p.tune_step[tix], p.tune_accepted[tix], p.tune_proposed[tix], p.tune_totproposed[tix], 0, 0.0, 0.0 };
    if (da) { tn.epsbar = p.da_epsbar[tix]; tn.hbar = p.da_hbar[tix]; }
    tn.phase = p.cnt ? (int)(tn.proposed % p.period) : 0;
    int sphase = kl.save_phase0;
    long long scol = kl.save_col0;
    double lt = cx.chain_ok ? p.LT[cx.chain] : 0.0;
    unsigned long long nacc = 0;
    bool stuck = false;                                  // slice sampler: step-out / shrink ran out of attempts
    const bool do_sum = p.sum != nullptr;
    long long held = do_sum ? p.held[cx.chain_ok ? cx.chain : 0] : 0;
    const __amdgpu_buffer_rsrc_t wX = mwin<NE>(cx, p.X, 0, p.D), wG = mwin<NE>(cx, p.GR, 0, p.D);

    // the committed state of the lane's chain (value, gradient) in registers; re-read after a rejected proposal, an accepted one stays where it is
    double xp[NE], gp[NE];
    {
        const int nv = cx.nv_here();
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const unsigned o = cx.off(e, nv);
            xp[e] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(wX, o, 0, 0));
            gp[e] = NEEDG ? __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(wG, o, 0, 0)) : 0.0;
        }
    }

    for (int s = 0; s < kl.nsteps; ++s) {
        const unsigned long long t = kl.t0 + (unsigned long long)s;
        if (p.cnt) tune_count_proposal(p, tn);
        bool acc = false;
        double ltp = lt;
        if constexpr (SAMPLER == KLARA_SAMPLER_HMC) {
            // iterate/HMC.jl:124-201, leapfrog! samplers.jl:122-134 in its merged fma form
            mnormals_lds<NE>(cx, p.seed, gchain, t, momw);                           // HMC.jl:135
            double k0[1] = { 0.0 };
            mom_read<NE>(momw, [&](int, double m) { k0[0] = k0[0] + m * m; });
            mreduce<1>(k0, cx.lane);
            const double H0 = lt - 0.5 * k0[0];                                      // HMC.jl:137
            const double eps = tn.step, halfe = 0.5 * eps;
            mom_update<NE>(momw, [&](int e, double m) { return kd_fma(halfe, gp[e], m); });
            // dual averaging: per-chain trip count (iterate/HMC.jl:142-144); the tile runs to its longest trajectory.  A finished chain's value and
            // momentum stop changing, so the later evaluations recompute for it the gradient and the log-target it already has, bit for bit.
            const int nl = da ? (cx.chain_ok ? da_nleaps(p, eps) : 1) : p.nleaps;    // (a padding lane must not set the wavefront's trip count)
            const int nlmax = da ? wave_max_int(nl) : nl;
            for (int l = 0; l < nlmax; ++l) {
                const bool go = !da || l < nl;
                mom_read<NE>(momw, [&](int e, double m) { const double v = kd_fma(eps, m, xp[e]); xp[e] = go ? v : xp[e]; });
                ltp = logitm_target<NE, true>(p, F, ypad, ldsL12, nblocks, cx.lane, xp, gp);       // samplers.jl:132 (the last one is also HMC.jl:157)
                const double kf = l + 1 < nl ? eps : halfe;
                mom_update<NE>(momw, [&](int e, double m) { const double v = kd_fma(kf, gp[e], m); return go ? v : m; });
            }
            double red[1] = { 0.0 };
            mom_read<NE>(momw, [&](int, double m) { red[0] = red[0] + m * m; });
            mreduce<1>(red, cx.lane);
            const double H1 = ltp - 0.5 * red[0];                                    // HMC.jl:159
            const double ratio = H1 - H0;                                            // HMC.jl:161
            const double ex = kd_exp(ratio);
            const double a = 1.0 < ex ? 1.0 : ex;                                    // HMC.jl:163
            const double u = kd_accept_uniform(kd_stream_block(p.seed, gchain, t, (uint32_t)((p.D + 1) >> 1)));
            acc = u < a;                                                             // HMC.jl:165
            if (da) da_update(p, tn, (long long)t + 1, a);                           // HMC.jl:225-249
        } else if constexpr (SAMPLER == KLARA_SAMPLER_MALA) {
            // iterate/MALA.jl:78-128: the proposal overwrites the value registers as its normals are consumed, the current value takes each
            // normal's place in the lane's LDS column (backward term, and what a rejecting lane goes back to)
            const double h = tn.step, halfh = 0.5 * h, sq = __builtin_sqrt(h), half_inv_h = 0.5 * (1.0 / h);
            double s1 = 0.0;
            mnormals_lds<NE>(cx, p.seed, gchain, t, momw);
#pragma unroll
            for (int e0 = 0; e0 < NE; e0 += 8) {
                double zz[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) zz[j] = momw[(e0 + j) * 64];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int e = e0 + j;
                    momw[e * 64] = xp[e];
                    const double mu = xp[e] + halfh * gp[e];                          // MALA.jl:83
                    xp[e] = mu + sq * zz[j];                                          // MALA.jl:84
                    const double q1 = mu - xp[e];
                    s1 = s1 + (q1 * q1) * half_inv_h;                                 // MALA.jl:90
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            ltp = logitm_target<NE, true>(p, F, ypad, ldsL12, nblocks, cx.lane, xp, gp);           // MALA.jl:86 (gp: the proposal's gradient from here on)
            double s2 = 0.0, red[2];
#pragma unroll
            for (int e0 = 0; e0 < NE; e0 += 8) {
                double xc[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) xc[j] = momw[(e0 + j) * 64];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int e = e0 + j;
                    const double mup = xp[e] + halfh * gp[e];                         // MALA.jl:91
                    const double q2 = mup - xc[j];
                    s2 = s2 + (q2 * q2) * half_inv_h;                                 // MALA.jl:92
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            red[0] = s1; red[1] = s2;
            mreduce<2>(red, cx.lane);
            double ratio = ltp - lt;                                                  // MALA.jl:88
            ratio += red[0];
            ratio -= red[1];
            acc = ratio > 0.0;                                                        // MALA.jl:94
            if (__any(!acc && ratio > KD_LOG_UMIN_GUARD)) {
                const double u = kd_accept_uniform(kd_stream_block(p.seed, gchain, t, (uint32_t)((p.D + 1) >> 1)));
                acc = acc || ratio > kd_log_u01(u);
            }
        } else if constexpr (SLICE) {
            // iterate/SliceSampler.jl:60-109 — the 16 chains of the tile out of lockstep (slice_dense_free, klara_dense.h): a probe is a full evaluation of the
            // log-target (:77-94) = pass 1 and the rows for the tile's 16 chains, each chain at its own coordinate and stage
            double cur = lt;
            slice_dense_free<NE>(p, cx, gchain, t, xp, cur, stuck, [&](const double (&xt)[NE]) {
                double gd[NE];
                return logitm_target<NE, false>(p, F, ypad, ldsL12, nblocks, cx.lane, xt, gd);
            });
            ltp = cur;
            acc = true;                                  // the slice sampler always moves (SliceSampler.jl:108)
        } else {
            // iterate/MH.jl:72-124: the log-target alone — pass 1 and the rows, no gradient pass
            mnormals_each<NE, KLARA_BIG_NORMALS_WAYS_MH>(cx, p.seed, gchain, t, [&](int e, double z) {
                const double sg = ldsSig[4 * e + cx.q];                               // (0 past D)
                momw[e * 64] = xp[e];
                xp[e] = xp[e] + sg * z;                                               // MH.jl:79
            });
            ltp = logitm_target<NE, false>(p, F, ypad, ldsL12, nblocks, cx.lane, xp, gp);          // MH.jl:81
            const double ratio = ltp - lt;                                            // MH.jl:83
            acc = ratio > 0.0;                                                        // MH.jl:97
            if (__any(!acc && ratio > KD_LOG_UMIN_GUARD)) {
                const double u = kd_accept_uniform(kd_stream_block(p.seed, gchain, t, (uint32_t)((p.D + 1) >> 1)));
                acc = acc || ratio > kd_log_u01(u);
            }
        }

        // fold, commit, reload — every lane takes every step under a wave-uniform condition (klara_dense_big.h: no divergent branch around the arrays)
        if (do_sum && __any(acc && held > 0)) {          // leaving a state after `held` saved steps: fold it into the sums
            const bool fold = acc && held > 0;
            const double hf = (double)held;
            const __amdgpu_buffer_rsrc_t ws = mwin<NE>(cx, p.sum, 0, p.D), wq = mwin<NE>(cx, p.sumsq, 0, p.D);
            const int nv = cx.nv_here();
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const unsigned o = fold ? cx.off(e, nv) : KLARA_BUF_OOB;
                const double xo = XMEM ? __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(wX, o, 0, 0)) : momw[e * 64];
                const double sv = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(ws, o, 0, 0));
                const double qv = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(wq, o, 0, 0));
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(kd_uint2, sv + hf * xo), ws, o, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(kd_uint2, qv + hf * (xo * xo)), wq, o, 0, 0);
            }
            held = fold ? 0 : held;
        }
        if (__any(acc)) {                                // commit (HMC.jl:166-176): what a later reject re-reads
            const int nv = cx.nv_here();
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const unsigned o = acc ? cx.off(e, nv) : KLARA_BUF_OOB;
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(kd_uint2, xp[e]), wX, o, 0, 0);
                if (NEEDG) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(kd_uint2, gp[e]), wG, o, 0, 0);
            }
        }
        lt = acc ? ltp : lt;
        if (__any(!acc)) {                               // a lane that rejected holds the proposal: back to the committed state
            const int nv = cx.nv_here();
#pragma unroll
            for (int e0 = 0; e0 < NE; e0 += 8) {
                double xc[8], gc[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const unsigned o = acc ? KLARA_BUF_OOB : cx.off(e0 + j, nv);
                    if (!XMEM) xc[j] = momw[(e0 + j) * 64];
                    else xc[j] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(wX, o, 0, 0));
                    if (NEEDG) gc[j] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(wG, o, 0, 0));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int e = e0 + j;
                    xp[e] = acc ? xp[e] : xc[j];
                    if (NEEDG) gp[e] = acc ? gp[e] : gc[j];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        nacc += acc ? 1ull : 0ull;
        tn.accepted += (p.cnt && acc && !SLICE) ? 1 : 0;          // (the slice sampler never counts accepts)
        if (accept_out != nullptr) {
            const __amdgpu_buffer_rsrc_t wa = __builtin_amdgcn_make_buffer_rsrc((void*)(accept_out + (long long)s * p.nchains + cx.first_chain), 0,
                                                                                __builtin_amdgcn_readfirstlane(cx.here), 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b8((unsigned char)(acc ? 1 : 0), wa, (cx.chain_ok && cx.q == 0) ? (unsigned)cx.cl : KLARA_BUF_OOB, 0, 0);
        }
        if (!p.pooled && !da) tuning_block_uniform(p, tn);
        else if (da && p.cnt && tn.phase == 0 && (long long)t + 1 <= p.da_nadapt) {     // verbose report block, iterate/HMC.jl:229-243
            tn.totproposed += tn.proposed; tn.accepted = 0; tn.proposed = 0;
        }
        const long long i1 = (long long)t + 1;
        const bool in_post = i1 > p.burnin && i1 <= p.nsteps_total;
        const bool save_now = in_post && sphase == 0;
        if (in_post) sphase = (sphase + 1 == (int)p.thinning) ? 0 : sphase + 1;
        if (save_now) {                                  // save rule (BasicMCJob.jl:226-231): the registers hold the committed state
            const long long col = scol++;
            if (do_sum) held += 1;
            if (col < p.hist_cols) {
                const int nv = cx.nv_here();
                if (p.hist != nullptr) {
                    const __amdgpu_buffer_rsrc_t wh = mwin<NE>(cx, p.hist, col * p.nchains, p.D);
#pragma unroll
                    for (int e = 0; e < NE; ++e) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(kd_uint2, xp[e]), wh, cx.off(e, nv), 0, 0);
                }
                if (NEEDG && p.hist_g != nullptr) {
                    const __amdgpu_buffer_rsrc_t wh = mwin<NE>(cx, p.hist_g, col * p.nchains, p.D);
#pragma unroll
                    for (int e = 0; e < NE; ++e) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(kd_uint2, gp[e]), wh, cx.off(e, nv), 0, 0);
                }
                if (p.hist_lt != nullptr) {
                    const __amdgpu_buffer_rsrc_t wl = mwin<NE>(cx, p.hist_lt, col * p.nchains, 1);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(kd_uint2, lt), wl,
                                                          (cx.chain_ok && cx.q == 0) ? (unsigned)cx.cl * 8u : KLARA_BUF_OOB, 0, 0);
                }
            }
        }
    }

    if (SLICE && stuck && cx.chain_ok && cx.q == 0) klara_raise(p.error_flag, KLARA_ERR_SLICE_STUCK);
    if (cx.chain_ok && cx.q == 0) {
        p.LT[cx.chain] = lt;
        p.naccept[cx.chain] += nacc;
        if (do_sum) p.held[cx.chain] = held;
        if (da) { p.da_epsbar[cx.chain] = tn.epsbar; p.da_hbar[cx.chain] = tn.hbar; }
        if (!p.pooled) {
            p.tune_step[cx.chain] = tn.step;
            p.tune_accepted[cx.chain] = tn.accepted;
            p.tune_proposed[cx.chain] = tn.proposed;
            p.tune_totproposed[cx.chain] = tn.totproposed;
        } else if (p.cnt) {
            atomicAdd((unsigned long long*)p.pooled_accepted, (unsigned long long)tn.accepted - (unsigned long long)p.tune_accepted[0]);
        }
    }
}

// initialize! (MALA.jl:76-90 / HMC.jl:106-120 / MH.jl:72-85): lt and the gradient at X, finiteness asserts
template <int NE>
__global__ __launch_bounds__(256) void k_logit_mfma_init(const KParams p, const double* __restrict__ F, const double* __restrict__ ypad_, int nblocks, int needgrad)
{
    __shared__ __attribute__((aligned(16))) double ldsL12[256];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) ldsL12[i] = kd_l12tab_dev[i];
    kd_tables_to_lds();
    const MfmaCtx<NE> cx = make_mctx<NE>(p);
    double x[NE], g[NE];
    mload<NE>(cx, p.X, p.D, x);
    const double lt = logitm_target<NE, true>(p, F, (const gdouble*)ypad_, ldsL12, nblocks, cx.lane, x, g);
    bool bad = cx.chain_ok && !kfinite(lt);
    if (needgrad) {
        const __amdgpu_buffer_rsrc_t wG = mwin<NE>(cx, p.GR, 0, p.D);
        const int nv = cx.nv_here();
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            bad = bad || (cx.chain_ok && !kfinite(g[e]));
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(kd_uint2, g[e]), wG, cx.off(e, nv), 0, 0);
        }
    }
    if (cx.chain_ok && cx.q == 0) p.LT[cx.chain] = lt;
    if (bad) klara_raise(p.error_flag, KLARA_ERR_NONFINITE_INIT);
}
