// klara_dense_split.h — the dense-Gaussian target beyond D = 256 on the FP64 matrix cores: a WORKGROUP carries the tile of 16 chains (layout kind 6).
//
// klara_dense_big.h gives one wavefront the whole vectors of its 16 chains: NE = 64 elements per lane at D = 256 is what 256 architectural + 256
// accumulator registers hold, at one wavefront per SIMD.  Beyond that neither the value nor the gradient fits a lane.  Here the W = ceil(D / 64)
// wavefronts of a workgroup SHARE the tile: wavefront w owns the 16 elements e = 16 w .. 16 w + 15 of every lane's column (lane (q, chain) of
// klara_dense.h: dimension i = 4 e + q), i.e. the rows 64 w .. 64 w + 63 of the gradient  G' = P X'  — four 16-row tiles of
// v_mfma_f64_16x16x4 — and the same rows of every element-wise update (normals, proposal, kicks, sums).  What a wavefront needs from the others is
// the B operand of its matrix pass, the whole (x - mu) of the 16 chains: every wavefront writes its 16 elements per lane to the LDS block
// xb[k-step][lane] (conflict-free 8-byte writes: 8 KB per wavefront), a barrier, and the pass reads one 512-byte row per k-step.  The A fragments
// come from the same k-major stream as klara_dense_big.h's, ((kk * MT + t) * 64 + lane), MT = 4 W: the four tiles of wavefront w are 2 KB in a row per
// k-step, taken through a ring of 8 buffer loads with scalar offsets.  The sums of a transition (x.g, the proposal terms, the kinetic energy)
// are lane partials over the lane's 16 elements in ascending order, the 4-lane tree (q0 + q1) + (q2 + q3) inside the wavefront, then the
// wavefronts' values in ascending order through LDS — the oracle's layout kind 6 (G = W, E = 16).
//
// A wavefront holds 16 elements per lane whatever D is: one kernel per sampler serves 257 <= D <= 1024 (W = 5 .. 16 wavefronts; the K loop and the
// wavefront count are run-time values), at 2 .. 4 wavefronts per SIMD — the vector work of one (Box-Muller: 85 % of MALA's vector instructions) runs
// under the matrix passes of the others, which the 512-register kernels of klara_dense_big.h cannot do.
// The state follows klara_dense.h's k_dense_transitions: registers hold the proposal, X / GR the committed state (written at every accept,
// re-read after a reject); MALA's backward term reads the current value from X.
// MH, MALA, HMC (every tuner, dual averaging with per-chain trip counts), every monitor of the dense layouts.
#pragma once
#include "klara_dense.h"

#define KLARA_SPLIT_NEW 16             // elements per lane and wavefront
#define KLARA_SPLIT_WMAX 16            // wavefronts per workgroup (D <= 1024)

struct SplitCtx {
    MfmaCtx<KLARA_SPLIT_NEW> m;        // the wavefront's 16 elements per lane as a 16-element column: offsets and validity shifted by 16 w
    int w, W;                          // this wavefront, wavefronts per workgroup (scalars)
    double* xb;                        // LDS: the tile's x - mu, [k-step = element][lane]  (+ two rows the last prefetch may touch)
    double* rbuf;                      // LDS: 2 x [value][wavefront][chain] partial sums (ping-pong)
    const double* ldsMu;               // LDS: mu[4 e + q] at [4 e + q], zero-padded (HASMU)
    __amdgpu_buffer_rsrc_t wP;         // the fragment stream
    unsigned par;                      // reduction parity
};

__device__ __forceinline__ SplitCtx make_sctx(const KParams& p, const double* Pfrag, char* smem, bool hasmu)
{
    SplitCtx s;
    MfmaCtx<KLARA_SPLIT_NEW>& c = s.m;
    s.W = (int)(blockDim.x >> 6);
    s.w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    c.lane = threadIdx.x & 63;
    c.q = c.lane >> 4;
    c.cl = c.lane & 15;
    c.first_chain = (long long)blockIdx.x * 16;
    const long long left = p.nchains - c.first_chain;
    c.here = left < 16 ? (left > 0 ? (int)left : 0) : 16;
    c.chain = c.first_chain + c.cl;
    c.chain_ok = c.cl < c.here;
    int nv = c.chain_ok ? (p.D - c.q + 3) / 4 - KLARA_SPLIT_NEW * s.w : 0;
    c.nv = nv < 0 ? 0 : (nv > KLARA_SPLIT_NEW ? KLARA_SPLIT_NEW : nv);
    c.voff0 = (unsigned)((c.cl * p.D + c.q) * 8 + 32 * KLARA_SPLIT_NEW * s.w);
    const int NE = KLARA_SPLIT_NEW * s.W;
    s.xb = reinterpret_cast<double*>(smem);
    s.rbuf = s.xb + (size_t)(NE + 2) * 64;
    s.ldsMu = s.rbuf + 2 * 3 * KLARA_SPLIT_WMAX * 16;
    const unsigned long long a = (unsigned long long)Pfrag;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    s.wP = __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0,
                                             __builtin_amdgcn_readfirstlane(4 * s.W * (NE + 2) * 512), 0x00020000);
    s.par = 0u;
    return s;
}
// LDS bytes of a workgroup of W wavefronts
static inline size_t klara_split_lds_bytes(int W, bool hasmu)
{
    const size_t NE = (size_t)KLARA_SPLIT_NEW * W;
    return sizeof(double) * ((NE + 2) * 64 + 2 * 3 * KLARA_SPLIT_WMAX * 16 + (hasmu ? 4 * NE : 0));
}

// acc[j] = tile 4 w + j of +P (x - mu), from zero: the K loop over the NE = 16 W rows of xb, two k-steps (8 fragments = the ring) per trip
__device__ __forceinline__ void split_pass(const SplitCtx& s, kd_double4 (&acc)[4])
{
    const int NE = KLARA_SPLIT_NEW * s.W;
    const unsigned strideK = (unsigned)(4 * s.W) * 512u;          // bytes between k-steps: MT fragments
    const unsigned voff = (unsigned)s.m.lane * 8u;
    unsigned soff = (unsigned)s.w * 2048u;                        // this wavefront's four tiles inside a k-step
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = (kd_double4){ 0.0, 0.0, 0.0, 0.0 };
    double ring[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
        ring[i] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(s.wP, voff, soff + (unsigned)(i >> 2) * strideK + (unsigned)(i & 3) * 512u, 0));
    const double* xl = s.xb + s.m.lane;
    double b0 = xl[0], b1 = xl[64];
    for (int k2 = 0; k2 < NE; k2 += 2) {
        soff += 2u * strideK;
        xl += 128;
        const double n0 = xl[0], n1 = xl[64];                     // (the last trip reads the two spare rows)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const double a = ring[i];
            // (the stream ends with two k-steps of zeros: no guard on the last trip — a scalar offset is not range-checked)
            ring[i] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(s.wP, voff, soff + (unsigned)(i >> 2) * strideK + (unsigned)(i & 3) * 512u, 0));
            acc[i & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, i < 4 ? b0 : b1, acc[i & 3], 0, 0, 0);
        }
        b0 = n0; b1 = n1;
        __builtin_amdgcn_sched_barrier(0);
    }
}

// g = -+P (x - mu) for the lane's 16 elements of the proposal x: every wavefront publishes its part of x - mu, then takes its rows of the product.
// PRE: a pass may still be reading xb (no barrier since the previous one: the leapfrog loop)
template <bool HASMU, bool NEG, bool PRE>
__device__ __forceinline__ void split_grad(const SplitCtx& s, const double (&x)[KLARA_SPLIT_NEW], double (&g)[KLARA_SPLIT_NEW])
{
    if (PRE) __syncthreads();
    double* const col = s.xb + (size_t)(KLARA_SPLIT_NEW * s.w) * 64 + s.m.lane;
    const double* const mu = s.ldsMu + 4 * KLARA_SPLIT_NEW * s.w + s.m.q;
#pragma unroll
    for (int e = 0; e < KLARA_SPLIT_NEW; ++e) col[e * 64] = HASMU ? x[e] - mu[4 * e] : x[e];
    __syncthreads();
    kd_double4 acc[4];
    split_pass(s, acc);
#pragma unroll
    for (int e = 0; e < KLARA_SPLIT_NEW; ++e) g[e] = NEG ? -acc[e >> 2][e & 3] : acc[e >> 2][e & 3];
}

// all-reduce of N sums over the tile's W wavefronts: 4-lane tree inside the wavefront, then the wavefronts in ascending order (oracle: layout kind 6)
template <int N>
__device__ __forceinline__ void split_reduce(SplitCtx& s, double (&v)[N])
{
    static_assert(N <= 3, "rbuf holds three values");
    mreduce<N>(v, s.m.lane);
    double* const rb = s.rbuf + (size_t)(s.par & 1u) * 3 * KLARA_SPLIT_WMAX * 16;
    s.par ^= 1u;
    if (s.m.q == 0) {
#pragma unroll
        for (int i = 0; i < N; ++i) rb[(i * KLARA_SPLIT_WMAX + s.w) * 16 + s.m.cl] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double t = rb[(i * KLARA_SPLIT_WMAX) * 16 + s.m.cl];
        for (int w2 = 1; w2 < s.W; ++w2) t = t + rb[(i * KLARA_SPLIT_WMAX + w2) * 16 + s.m.cl];
        v[i] = t;
    }
}

// normals of the lane's 16 elements (mnormals of klara_dense.h with the element index 16 w + e: the pair's block slot moves by 16 w)
__device__ __forceinline__ void split_normals(const SplitCtx& s, unsigned long long seed, unsigned long long gchain, unsigned long long t,
                                              double (&z)[KLARA_SPLIT_NEW])
{
    const MfmaCtx<KLARA_SPLIT_NEW>& c = s.m;
    const uint32_t sh = (uint32_t)(c.q >> 1);
    const bool odd = (c.q & 1) != 0;
    const int nv = c.nv_here();
    const uint32_t lane_slot = (odd ? 2u : 0u) + sh + 16u * (uint32_t)s.w;
    MPairStash st = { { 0u, 0u, 0u, 0u } };
#pragma unroll
    for (int e = 0; e + 1 < KLARA_SPLIT_NEW; e += 2) {
        double z0, z1;
        mpair_normals(seed, gchain, t, e, lane_slot, st, z0, z1);
        const double recv = bperm_xor(odd ? z0 : z1, c.lane, 16);
        z[e] = e < nv ? (odd ? recv : z0) : 0.0;
        z[e + 1] = e + 1 < nv ? (odd ? z1 : recv) : 0.0;
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int SAMPLER, bool DA, bool HASMU, int WB>
__global__ __launch_bounds__(64 * WB)
void k_dense_split(const KParams* __restrict__ pp, const KLaunch kl, const double* __restrict__ Pfrag)
{
    static_assert(SAMPLER == KLARA_SAMPLER_HMC || SAMPLER == KLARA_SAMPLER_MALA || SAMPLER == KLARA_SAMPLER_MH, "HMC, MALA, MH");
    static_assert(!DA || SAMPLER == KLARA_SAMPLER_HMC, "dual averaging is wired into HMC only (HMC.jl:124-133)");
    constexpr int NE = KLARA_SPLIT_NEW;
    const KParams& p = *pp;
    guchar* const accept_out = p.accept != nullptr ? p.accept + kl.t0 * (unsigned long long)p.nchains : nullptr;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    SplitCtx sc = make_sctx(p, Pfrag, smem, HASMU);
    const MfmaCtx<NE>& cx = sc.m;
    {
        const int NEt = NE * sc.W;
        double* const muW = const_cast<double*>(sc.ldsMu);
        if (HASMU) { for (int i = threadIdx.x; i < 4 * NEt; i += blockDim.x) muW[i] = Pfrag[(size_t)4 * sc.W * (NEt + 2) * 64 + i]; }
        for (int i = threadIdx.x; i < 128; i += blockDim.x) sc.xb[(size_t)NEt * 64 + i] = 0.0;        // the spare rows
    }
    kd_tables_to_lds();          // (also the barrier for mu)
    const bool w0 = sc.w == 0;
    const auto dx = [&](const double (&v)[NE], int e) { return HASMU ? v[e] - sc.ldsMu[4 * (NE * sc.w + e) + cx.q] : v[e]; };
    const auto gch = [&]() { return (unsigned long long)(p.chain_offset + cx.chain_here()); };       // global chain id: the Philox subsequence
    const long long tix = p.pooled ? 0 : (cx.chain_ok ? cx.chain : 0);
    constexpr bool da = DA;
    const bool cnt = p.cnt != 0;
    // (every wavefront of the tile carries the tuner state, the log-target and the sojourn count of its lanes' chains: the same values, the same updates)
    TuneRegs tn = { p.tune_step[tix], p.tune_accepted[tix], p.tune_proposed[tix], p.tune_totproposed[tix], 0, 0.0, 0.0 };
    if (da) { tn.epsbar = p.da_epsbar[tix]; tn.hbar = p.da_hbar[tix]; }
    tn.phase = cnt ? (int)(tn.proposed % p.period) : 0;
    int sphase = kl.save_phase0;
    long long scol = kl.save_col0;
    double lt = cx.chain_ok ? p.LT[cx.chain] : 0.0;
    unsigned long long nacc = 0;
    const bool do_sum = p.sum != nullptr;
    long long held = do_sum ? p.held[cx.chain_ok ? cx.chain : 0] : 0;        // running sums in sojourn form (KParams::held)

    double xp[NE], gp[NE];
    bool have = false;           // the registers hold the committed state (the last proposal was accepted)
    for (int s = 0; s < kl.nsteps; ++s) {
        const unsigned long long t = kl.t0 + (unsigned long long)s;
        if (cnt) tune_count_proposal(p, tn);
        bool acc = false;
        double ltp = lt;
        if (!have) mload<NE>(cx, p.X, p.D, xp);                        // current value

        if constexpr (SAMPLER == KLARA_SAMPLER_HMC) {
            // iterate/HMC.jl:124-201, leapfrog! samplers.jl:122-134 (merged fma form: DESIGN.md section 2 (7))
            double mom[NE], red[2];
            if (!have) mload<NE>(cx, p.GR, p.D, gp);                   // HMC.jl:140
            split_normals(sc, p.seed, gch(), t, mom);                  // HMC.jl:135
            double k0[1] = { 0.0 };
#pragma unroll
            for (int e = 0; e < NE; ++e) k0[0] = k0[0] + mom[e] * mom[e];
            split_reduce<1>(sc, k0);
            const double H0 = lt - 0.5 * k0[0];                        // HMC.jl:137
            const double eps = tn.step, halfe = 0.5 * eps;
#pragma unroll
            for (int e = 0; e < NE; ++e) mom[e] = kd_fma(halfe, gp[e], mom[e]);
            // dual averaging: per-chain trip count (iterate/HMC.jl:142-144); the tile runs to its longest trajectory (the same count in every
            // wavefront: each holds all 16 chains), a finished chain's lanes keep their state
            const int nl = da ? (cx.chain_ok ? da_nleaps(p, eps) : 1) : p.nleaps;
            for (int l = 0; da ? __any(l < nl) : (l < nl); ++l) {
                const bool go = !da || l < nl;
#pragma unroll
                for (int e = 0; e < NE; ++e) { const double v = kd_fma(eps, mom[e], xp[e]); xp[e] = go ? v : xp[e]; }
                double gn[NE];
                split_grad<HASMU, false, true>(sc, xp, gn);
                const double nkf = l + 1 < nl ? -eps : -halfe;
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    const double v = kd_fma(nkf, gn[e], mom[e]);
                    mom[e] = go ? v : mom[e];
                    gp[e] = go ? -gn[e] : gp[e];
                }
            }
            double l1 = 0.0, k1 = 0.0;
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                l1 = l1 + dx(xp, e) * gp[e];                           // lt' = c + 1/2 (x'-mu).g'   (HMC.jl:157)
                k1 = k1 + mom[e] * mom[e];
            }
            red[0] = l1; red[1] = k1;
            split_reduce<2>(sc, red);
            ltp = p.gconst + 0.5 * red[0];
            const double H1 = ltp - 0.5 * red[1];                      // HMC.jl:159
            const double ratio = H1 - H0;                              // HMC.jl:161
            const double ex = kd_exp(ratio);
            const double a = 1.0 < ex ? 1.0 : ex;                      // HMC.jl:163
            const double u = kd_accept_uniform(kd_stream_block(p.seed, gch(), t, (uint32_t)((p.D + 1) >> 1)));
            acc = u < a;                                               // HMC.jl:165
            if (da) da_update(p, tn, (long long)t + 1, a);             // HMC.jl:225-249
        } else if constexpr (SAMPLER == KLARA_SAMPLER_MALA) {
            // iterate/MALA.jl:78-128
            double red[3];
            const double h = tn.step, halfh = 0.5 * h, sq = __builtin_sqrt(h), inv_h = 1.0 / h, half_inv_h = 0.5 * inv_h;
            if (!have) mload<NE>(cx, p.GR, p.D, gp);
            double s1 = 0.0;
            {
                double z[NE];
                split_normals(sc, p.seed, gch(), t, z);
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    const double mu = xp[e] + halfh * gp[e];           // MALA.jl:83
                    xp[e] = mu + sq * z[e];                            // MALA.jl:84
                    const double q1 = mu - xp[e];
                    s1 = s1 + (q1 * q1) * half_inv_h;                  // MALA.jl:90
                }
            }
            split_grad<HASMU, true, false>(sc, xp, gp);                // MALA.jl:86
            double l1 = 0.0, s2 = 0.0;
            {
                double xc[NE];
                mload<NE>(cx, p.X, p.D, xc);                           // the current value (X holds the committed state)
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    l1 = l1 + dx(xp, e) * gp[e];
                    const double mup = xp[e] + halfh * gp[e];          // MALA.jl:91
                    const double q2 = mup - xc[e];
                    s2 = s2 + (q2 * q2) * half_inv_h;                  // MALA.jl:92
                }
            }
            red[0] = l1; red[1] = s1; red[2] = s2;
            split_reduce<3>(sc, red);
            ltp = p.gconst + 0.5 * red[0];
            double ratio = ltp - lt;                                   // MALA.jl:88
            ratio += red[1];
            ratio -= red[2];
            acc = ratio > 0.0;                                         // MALA.jl:94
            if (!acc && ratio > KD_LOG_UMIN_GUARD) {
                const double u = kd_accept_uniform(kd_stream_block(p.seed, gch(), t, (uint32_t)((p.D + 1) >> 1)));
                acc = ratio > kd_log_u01(u);
            }
        } else {
            // iterate/MH.jl:72-124
            double red[1];
            {
                double z[NE];
                split_normals(sc, p.seed, gch(), t, z);
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    const int i = 4 * (NE * sc.w + e) + cx.q;
                    const double sg = i < p.D ? p.vecparam[i] : 0.0;
                    xp[e] = xp[e] + sg * z[e];                         // MH.jl:79
                }
            }
            split_grad<HASMU, true, false>(sc, xp, gp);                // MH.jl:81
            double l1 = 0.0;
#pragma unroll
            for (int e = 0; e < NE; ++e) l1 = l1 + dx(xp, e) * gp[e];
            red[0] = l1;
            split_reduce<1>(sc, red);
            ltp = p.gconst + 0.5 * red[0];
            const double ratio = ltp - lt;                             // MH.jl:83
            acc = ratio > 0.0;                                         // MH.jl:97
            if (!acc && ratio > KD_LOG_UMIN_GUARD) {
                const double u = kd_accept_uniform(kd_stream_block(p.seed, gch(), t, (uint32_t)((p.D + 1) >> 1)));
                acc = ratio > kd_log_u01(u);
            }
        }

        if (do_sum && __any(acc && held > 0)) {          // leaving a state after `held` saved steps: fold it into the sums
            if (acc && held > 0) {
                const double hf = (double)held;
                double xo[NE];
                mload<NE>(cx, p.X, p.D, xo);
                const __amdgpu_buffer_rsrc_t ws = mwin<NE>(cx, p.sum, 0, p.D), wq = mwin<NE>(cx, p.sumsq, 0, p.D);
                const int nv = cx.nv_here();
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    const unsigned o = cx.off_fresh(e, nv);
                    const double sv = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(ws, o, 0, 0));
                    const double qv = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(wq, o, 0, 0));
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(kd_uint2, sv + hf * xo[e]), ws, o, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(kd_uint2, qv + hf * (xo[e] * xo[e])), wq, o, 0, 0);
                }
                held = 0;
            }
        }
        if (acc) {
            mstore<NE>(cx, p.X, p.D, xp);
            if (SAMPLER != KLARA_SAMPLER_MH) mstore<NE>(cx, p.GR, p.D, gp);
            lt = ltp;
        }
        have = acc;                                      // (a rejected proposal leaves the registers holding the proposal: re-read next time)
        nacc += acc ? 1ull : 0ull;
        if (cnt && acc) tn.accepted += 1;
        if (accept_out != nullptr && w0 && cx.chain_ok && cx.q == 0)
            accept_out[(long long)s * p.nchains + cx.chain_here()] = acc ? 1 : 0;
        if (!p.pooled && !da) tuning_block(p, tn);
        else if (da && cnt && tn.phase == 0 && (long long)t + 1 <= p.da_nadapt) {
            tn.totproposed += tn.proposed; tn.accepted = 0; tn.proposed = 0;
        }
        const long long i1 = (long long)t + 1;
        const bool in_post = i1 > p.burnin && i1 <= p.nsteps_total;
        const bool save_now = in_post && sphase == 0;
        if (in_post) sphase = (sphase + 1 == (int)p.thinning) ? 0 : sphase + 1;
        if (save_now) {
            const long long col = scol++;
            if (do_sum) held += 1;
            if (p.hist != nullptr) {
                double xs[NE];
                if (acc) {
#pragma unroll
                    for (int e = 0; e < NE; ++e) xs[e] = xp[e];
                } else {
                    mload<NE>(cx, p.X, p.D, xs);
                }
                if (col < p.hist_cols) mstore<NE>(cx, p.hist, p.D, xs, col * p.nchains);
            }
            if (p.hist_lt != nullptr && col < p.hist_cols && w0 && cx.chain_ok && cx.q == 0)
                p.hist_lt[col * p.nchains + cx.chain_here()] = lt;
            if (SAMPLER != KLARA_SAMPLER_MH && p.hist_g != nullptr && col < p.hist_cols) {
                double gs[NE];
                if (acc) {
#pragma unroll
                    for (int e = 0; e < NE; ++e) gs[e] = gp[e];
                } else {
                    mload<NE>(cx, p.GR, p.D, gs);
                }
                mstore<NE>(cx, p.hist_g, p.D, gs, col * p.nchains);
            }
        }
        // (a chain's X / GR rows are written by all W wavefronts and read back by them after a reject, each its own elements: no hazard across wavefronts)
    }

    if (w0 && cx.lane < cx.here) {                                       // q == 0 (lanes 0..15) on an existing chain, first wavefront
        const long long chain_e = cx.first_chain + cx.lane;
        p.LT[chain_e] = lt;
        p.naccept[chain_e] += nacc;
        if (do_sum) p.held[chain_e] = held;
        if (da) { p.da_epsbar[chain_e] = tn.epsbar; p.da_hbar[chain_e] = tn.hbar; }
        if (!p.pooled) {
            p.tune_step[chain_e] = tn.step;
            p.tune_accepted[chain_e] = tn.accepted;
            p.tune_proposed[chain_e] = tn.proposed;
            p.tune_totproposed[chain_e] = tn.totproposed;
        } else if (cnt) {
            atomicAdd((unsigned long long*)p.pooled_accepted, (unsigned long long)tn.accepted - (unsigned long long)p.tune_accepted[0]);
        }
    }
}

// initialize! for the dense target on the split layout: g = -P (x - mu), lt = c + 1/2 (x - mu).g, finiteness asserts
template <bool HASMU>
__global__ __launch_bounds__(1024) void k_dense_split_init(const KParams p, const double* __restrict__ Pfrag, int needgrad)
{
    constexpr int NE = KLARA_SPLIT_NEW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    SplitCtx sc = make_sctx(p, Pfrag, smem, HASMU);
    const MfmaCtx<NE>& cx = sc.m;
    {
        const int NEt = NE * sc.W;
        double* const muW = const_cast<double*>(sc.ldsMu);
        if (HASMU) { for (int i = threadIdx.x; i < 4 * NEt; i += blockDim.x) muW[i] = Pfrag[(size_t)4 * sc.W * (NEt + 2) * 64 + i]; }
        for (int i = threadIdx.x; i < 128; i += blockDim.x) sc.xb[(size_t)NEt * 64 + i] = 0.0;
    }
    __syncthreads();
    double x[NE], g[NE], red[1];
    mload<NE>(cx, p.X, p.D, x);
    split_grad<HASMU, true, false>(sc, x, g);
    double l1 = 0.0;
    bool bad = false;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        l1 = l1 + (HASMU ? x[e] - sc.ldsMu[4 * (NE * sc.w + e) + cx.q] : x[e]) * g[e];
        if (needgrad) bad = bad || !kfinite(g[e]);
    }
    red[0] = l1;
    split_reduce<1>(sc, red);
    const double lt = p.gconst + 0.5 * red[0];
    bad = bad || (cx.chain_ok && !kfinite(lt));
    if (needgrad) mstore<NE>(cx, p.GR, p.D, g);
    if (sc.w == 0 && cx.chain_ok && cx.q == 0) p.LT[cx.chain] = lt;
    if (bad) klara_raise(p.error_flag, KLARA_ERR_NONFINITE_INIT);
}
