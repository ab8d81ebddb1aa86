// klara_dense_big.h — HMC on the dense-Gaussian target beyond D = 128 on the FP64 matrix cores (NE = 8 ceil(D / 32) = 40, 48, 56, 64 elements per lane).
//
// What changes against klara_dense.h (D <= 128: P in LDS, three NE-element vectors per lane in 256 registers, 2 wavefronts per SIMD):
//   * P no longer fits the LDS (D = 256: 512 KB).  The A fragments are STREAMED from memory in the order of consumption — fragment array in
//     k-major order, ((kk * MT + t) * 64 + lane): consecutive 512-byte lines — through a ring of KLARA_DENSE_RING fragments per lane: the load
//     of step s + RING is issued when step s is consumed, ~RING x 64 MFMA cycles ahead of its use.  Every wavefront of every CU reads P in the
//     same order, so it lives in the L2s and the wavefronts of a CU share its lines in the L1.  No LDS staging, no barriers.
//   * A lane's vectors no longer fit 256 registers, and a 512-register wavefront has 256 ARCHITECTURAL + 256 ACCUMULATOR registers, of which
//     only MFMA operands / results can use the second half.  So: ONE wavefront per SIMD (workgroups of 4); the value x in architectural
//     registers (it is the MFMA's B operand and the target of the drift), the gradient P x in the accumulators (it IS the MFMA result), and the
//     MOMENTUM IN LDS — one private column per lane, momw[e * 64] —, touched twice per leapfrog (64 ds_read + 64 ds_write per lane against
//     1,024 MFMAs).
//   * HMC (Vanilla / AcceptanceRate tuners per chain or pooled, and dual averaging with its per-chain trip counts; every monitor), and MALA and MH:
//     their proposal overwrites the value registers, the current gradient (accumulators, kept from transition to transition) is consumed by the same
//     pass, and the CURRENT VALUE waits in the lane's LDS column (round 5; round 4 re-read it from X) for MALA's backward term and for a lane that
//     rejects — so nothing inside a launch reads X, which is written once, after the last transition; of the state only MALA's gradient goes to
//     memory per accepted transition (nt store: it is re-read after a reject only, and must not push P out of the L2).
//   * The slice sampler (round 5): a probe is a full evaluation = one pass over P for the tile's 16 chains, each chain at its own coordinate and
//     stage (slice_dense_free, klara_dense.h); no LDS column.
// Same MFMA instruction, same k-ascending fma chain per output (zero-padded rows / columns add exact zeros), same merged fma leapfrog, same
// 4-lane reduction tree: the oracle's ko_hmc / ko_dense_grad in layout kind 1, bit for bit.
#pragma once
#include "klara_dense.h"

#ifndef KLARA_DENSE_RING
#define KLARA_DENSE_RING 8            // fragments in flight per lane (NE >= 48; at NE = 64 a ring of 12 / 16 costs HMC 5 % / 8 %: registers)
#endif
#ifndef KLARA_DENSE_RING_SMALL
#define KLARA_DENSE_RING_SMALL 12     // ... at NE = 40, where the registers are there (HMC at D = 160: 58.7 -> 59.9 TFLOP/s)
#endif
// phase boundary of a transition: the scheduler may not move instructions across it (live ranges of one phase stay out of the next)
#ifndef KLARA_BIG_NO_PHASES
#define KLARA_BIG_PHASE() __builtin_amdgcn_sched_barrier(0)
#else
#define KLARA_BIG_PHASE() ((void)0)
#endif
#ifndef KLARA_BIG_NORMALS_GROUP
#define KLARA_BIG_NORMALS_GROUP 1     // Box-Muller pairs the scheduler may interleave (one wavefront per SIMD: nothing else hides a pair's dependent chains)
#endif
#ifndef KLARA_BIG_NORMALS_WAYS_MH
#define KLARA_BIG_NORMALS_WAYS_MH 4   // Box-Muller evaluations written side by side in MH's proposal (measured at D = 256: 1 -> 52.5, 2 -> 54.0, 4 -> 54.3 TFLOP/s)
#endif
#ifndef KLARA_BIG_NORMALS_WAYS
#define KLARA_BIG_NORMALS_WAYS 1      // ... in the normals HMC / MALA draw into the LDS column (MALA: 1 -> 41.5, 2 -> 35.5 (512 B of scratch), 4 -> 41.8; HMC: 4 -> -1.5 %)
#endif
#ifndef KLARA_BIG_RELOAD_CHUNK
#define KLARA_BIG_RELOAD_CHUNK 8      // elements re-read per group after a rejected proposal
#endif
#ifndef KLARA_BIG_X_EVERY
#define KLARA_BIG_X_EVERY 0           // 1: MALA / MH store an accepted value to X at once (round 4); 0: X is written once, when the launch ends
#endif
#ifndef KLARA_BIG_G_AUX
#define KLARA_BIG_G_AUX 2             // cache-policy bits of MALA's per-transition GR store: 2 = nt (measured: 0 -> 37.4, nt 41.4, sc1 nt 33.3 TFLOP/s at D = 256)
#endif
#ifndef KLARA_BIG_HMC_AUX
#define KLARA_BIG_HMC_AUX 0           // ... of HMC's X / GR stores
#endif
#ifndef KLARA_BIG_XC_CHUNK
#define KLARA_BIG_XC_CHUNK 8          // MALA's backward term: elements of the current value re-read per group (a power of two <= 8)
#endif

// acc[t] (tile t: elements 4t .. 4t+3 of the lane) = +P (x - mu), from zero
template <int NE, bool HASMU>
__device__ __forceinline__ void dense_stream(const double* __restrict__ gP, int lane, const double (&x)[NE], kd_double4 (&acc)[NE / 4],
                                             const double* ldsMu)
{
    constexpr int MT = NE / 4, S = NE * MT, RING = NE <= 40 ? KLARA_DENSE_RING_SMALL : KLARA_DENSE_RING;
    static_assert(NE % 4 == 0 && S > RING, "whole 16-row tiles");
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = (kd_double4){ 0.0, 0.0, 0.0, 0.0 };
    // (P does not change during a launch, so the compiler would hoist all S fragment loads out of the leapfrog loop — 1,024 doubles per lane,
    // i.e. into scratch; behind the empty asm the address is "new" in every call and the loads stay where they are written)
    // (kept in the global address space: global loads return in order, so the ring is waited for one fragment at a time — vmcnt(RING - 1))
    const gdouble* src = (const gdouble*)gP + lane;
    __asm__ volatile("" : "+v"(src));
    double ring[RING];
#pragma unroll
    for (int i = 0; i < RING; ++i) ring[i] = src[(size_t)i * 64];
#pragma unroll
    for (int kk = 0; kk < NE; ++kk) {
        const double b = HASMU ? x[kk] - ldsMu[4 * kk + (lane >> 4)] : x[kk];
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const int s = kk * MT + t;
            const double a = ring[s % RING];
            if (s + RING < S) ring[s % RING] = src[(size_t)(s + RING) * 64];
            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// the momentum draw of mnormals (klara_dense.h: the even / odd lanes of a chain evaluate alternate Box-Muller pairs and swap halves),
// written to the lane's LDS column
// ... and the same draw handed to a callback, element by element (f(e, z_e)): MALA / MH form their proposal from it where it is drawn
template <int NE, int W, class F>
__device__ __forceinline__ void mnormals_each(const MfmaCtx<NE>& c, unsigned long long seed, unsigned long long gchain, unsigned long long t, F f)
{
    static_assert(NE % 8 == 0, "pairs of elements, and every block's second pair in the same lane");
    const uint32_t sh = (uint32_t)(c.q >> 1);
    const bool odd = (c.q & 1) != 0;
    const int nv = c.nv_here();
    const uint32_t lane_slot = (odd ? 2u : 0u) + sh;
    if constexpr (W == 1) {
    MPairStash st = { { 0u, 0u, 0u, 0u } };
#pragma unroll
    for (int e = 0; e + 1 < NE; e += 2) {
        double z0, z1;
        mpair_normals(seed, gchain, t, e, lane_slot, st, z0, z1);
        const double recv = bperm_xor(odd ? z0 : z1, c.lane, 16);
        f(e, e < nv ? (odd ? recv : z0) : 0.0);
        f(e + 1, e + 1 < nv ? (odd ? z1 : recv) : 0.0);
        if ((e / 2 + 1) % KLARA_BIG_NORMALS_GROUP == 0) __builtin_amdgcn_sched_barrier(0);
    }
    } else {
    // Round 5: the Box-Muller evaluations of loop slots e and e + 2 (W = 2; 4: e .. e + 6) side by side, statement by statement (mpair_normals_n):
    // this wavefront is alone on its SIMD, where a fully dependent instruction stream issues every 10.7 cycles and four independent ones every 5.5
    // (profiles/r3_ubench_issue_cadence.txt).  Slots e, e + 2 with (e >> 2) even form a Philox block each and leave its words (z, w) to slots
    // e + 4, e + 6 (mpair_normals: one block per two slots).
    static_assert(W == 2 || W == 4, "1, 2 or 4 evaluations side by side");
    uint32_t stash[2][2] = { { 0u, 0u }, { 0u, 0u } };
#pragma unroll
    for (int e0 = 0; e0 < NE; e0 += 2 * W) {
        uint32_t wa[W], wb[W];
        double z0[W], z1[W];
#pragma unroll
        for (int j = 0; j < W; ++j) {
            const int e = e0 + 2 * j, si = (e >> 1) & 1;
            if ((e >> 2) & 1) { wa[j] = stash[si][0]; wb[j] = stash[si][1]; }
            else {
                const uint32_t base = ((2u * (uint32_t)e) & 7u) | (((2u * (uint32_t)e) >> 4) << 3);
                const kd_u32x4 b = kd_stream_block(seed, gchain, t, base + lane_slot);
                wa[j] = b.x; wb[j] = b.y; stash[si][0] = b.z; stash[si][1] = b.w;
            }
        }
        mpair_normals_n<W>(wa, wb, z0, z1);
#pragma unroll
        for (int j = 0; j < W; ++j) {
            const int e = e0 + 2 * j;
            const double recv = bperm_xor(odd ? z0[j] : z1[j], c.lane, 16);
            f(e, e < nv ? (odd ? recv : z0[j]) : 0.0);
            f(e + 1, e + 1 < nv ? (odd ? z1[j] : recv) : 0.0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    }
}

template <int NE>
__device__ __forceinline__ void mnormals_lds(const MfmaCtx<NE>& c, unsigned long long seed, unsigned long long gchain, unsigned long long t,
                                             double* momw)
{
    mnormals_each<NE, KLARA_BIG_NORMALS_WAYS>(c, seed, gchain, t, [&](int e, double z) { momw[e * 64] = z; });
}

// the lane's momentum column in LDS, touched in chunks of 8 elements: 8 reads in flight, then the 8 updates (one ds_read per element and a
// wait after each costs the LDS latency 64 times per pass)
#define KLARA_MOM_CHUNK 8
template <int NE, class F>
__device__ __forceinline__ void mom_read(const double* momw, F f)
{
    static_assert(NE % KLARA_MOM_CHUNK == 0, "whole chunks");
#pragma unroll
    for (int e0 = 0; e0 < NE; e0 += KLARA_MOM_CHUNK) {
        double m[KLARA_MOM_CHUNK];
#pragma unroll
        for (int j = 0; j < KLARA_MOM_CHUNK; ++j) m[j] = momw[(e0 + j) * 64];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < KLARA_MOM_CHUNK; ++j) f(e0 + j, m[j]);
        __builtin_amdgcn_sched_barrier(0);
    }
}
template <int NE, class F>
__device__ __forceinline__ void mom_update(double* momw, F f)       // m = f(e, m)
{
#pragma unroll
    for (int e0 = 0; e0 < NE; e0 += KLARA_MOM_CHUNK) {
        double m[KLARA_MOM_CHUNK];
#pragma unroll
        for (int j = 0; j < KLARA_MOM_CHUNK; ++j) m[j] = momw[(e0 + j) * 64];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < KLARA_MOM_CHUNK; ++j) momw[(e0 + j) * 64] = f(e0 + j, m[j]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int SAMPLER, int NE, bool HASMU = false, bool DA = false>
__global__ __launch_bounds__(256)
void k_dense_big(const KParams* __restrict__ pp, const KLaunch kl, const double* __restrict__ Pfrag)
{
    static_assert(SAMPLER == KLARA_SAMPLER_HMC || SAMPLER == KLARA_SAMPLER_MALA || SAMPLER == KLARA_SAMPLER_MH || SAMPLER == KLARA_SAMPLER_SLICE, "HMC, MALA, MH, slice");
    constexpr bool SLICE = SAMPLER == KLARA_SAMPLER_SLICE;
    constexpr bool NEEDG = SAMPLER != KLARA_SAMPLER_MH && !SLICE;    // MH and the slice sampler carry no gradient (GR is not written)
    constexpr bool KEEPG = NEEDG;                               // the committed gradient stays in the accumulators between transitions (re-read from GR only after a reject)
    constexpr bool XLDS = SAMPLER != KLARA_SAMPLER_HMC && !SLICE;   // MALA / MH: the current value waits in the lane's LDS column (MALA: where its normals were, for the backward term) — a rejected proposal is undone from there
    // ... so nothing inside a launch reads X: the value is written ONCE, after the last transition (it was 2 KB per chain and accepted transition at D = 256:
    // half of MALA's and all of MH's write stream, 4.3 GB per 32-transition launch, through the L2 that also holds P)
    constexpr bool XONCE = XLDS && !KLARA_BIG_X_EVERY;
    static_assert(!DA || SAMPLER == KLARA_SAMPLER_HMC, "dual averaging is wired into HMC only (HMC.jl:124-133)");
    constexpr bool da = DA;
    const KParams& p = *pp;
    guchar* const accept_out = p.accept != nullptr ? p.accept + kl.t0 * (unsigned long long)p.nchains : nullptr;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MT = NE / 4;
    double* const ldsMuW = reinterpret_cast<double*>(smem);                     // mu[4 e + q] at [4 e + q], zero-padded (HASMU only)
    if (HASMU) { for (int i = threadIdx.x; i < 4 * NE; i += blockDim.x) ldsMuW[i] = Pfrag[(size_t)MT * NE * 64 + i]; }
    const double* const ldsMu = ldsMuW;
    // MH's proposal scales, sigma[4 e + q] at [4 e + q], 0 past D (round 5: a buffer load per element inside the normals' loop was an exposed memory
    // round trip per Box-Muller pair at one wavefront per SIMD)
    constexpr bool SIGLDS = SAMPLER == KLARA_SAMPLER_MH;
    double* const ldsSig = ldsMuW + (HASMU ? 4 * NE : 0);
    if (SIGLDS) { for (int i = threadIdx.x; i < 4 * NE; i += blockDim.x) ldsSig[i] = (p.vecparam != nullptr && i < p.D) ? p.vecparam[i] : 0.0; }
    kd_tables_to_lds();          // (also the barrier for mu / sigma)
    const MfmaCtx<NE> cx = make_mctx<NE>(p);
    double* const momw = ldsSig + (SIGLDS ? 4 * NE : 0) + (SLICE ? 0 : (size_t)(threadIdx.x >> 6) * NE * 64 + cx.lane);     // (the slice sampler has none) this lane's column: momentum (HMC), normals then the current value (MALA), the current value (MH)
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

    // the committed state of the lane's chain: value in registers, gradient in the accumulator tiles (element e = ga[e >> 2][e & 3]).  They are
    // (re)read from memory at the start of the launch and after a rejected proposal; an accepted proposal simply stays where it is.
    double xp[NE];
    kd_double4 ga[MT];
    const auto reload = [&]() {
        const int nv = cx.nv_here();
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const unsigned o = cx.off(e, nv);
            xp[e] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(wX, o, 0, 0));
            if (KEEPG) ga[e >> 2][e & 3] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(wG, o, 0, 0));
        }
    };
    reload();

    for (int s = 0; s < kl.nsteps; ++s) {
        const unsigned long long t = kl.t0 + (unsigned long long)s;
        if (p.cnt) tune_count_proposal(p, tn);
        bool acc = false;
        double ltp = lt;
        if constexpr (SAMPLER == KLARA_SAMPLER_HMC) {
            // iterate/HMC.jl:124-201, leapfrog! samplers.jl:122-134 (merged fma form: DESIGN.md section 2 (7))
            mnormals_lds<NE>(cx, p.seed, gchain, t, momw);                           // HMC.jl:135
            double k0[1] = { 0.0 };
            mom_read<NE>(momw, [&](int, double m) { k0[0] = k0[0] + m * m; });
            mreduce<1>(k0, cx.lane);
            const double H0 = lt - 0.5 * k0[0];                                      // HMC.jl:137
            const double eps = tn.step, halfe = 0.5 * eps;
            mom_update<NE>(momw, [&](int e, double m) { return kd_fma(halfe, (double)ga[e >> 2][e & 3], m); });
            // dual averaging: per-chain trip count (iterate/HMC.jl:142-144); the 16 chains of the tile run to the longest trajectory.  A finished
            // chain's value and momentum stop changing, so the gradient the later passes recompute for it is the one it already has, bit for bit:
            // only the two updates are masked, the accumulators need no second copy.
            const int nl = da ? (cx.chain_ok ? da_nleaps(p, eps) : 1) : p.nleaps;        // (a padding lane must not set the wavefront's trip count)
            const int nlmax = da ? wave_max_int(nl) : nl;
            for (int l = 0; l < nlmax; ++l) {
                const bool go = !da || l < nl;
                mom_read<NE>(momw, [&](int e, double m) { const double v = kd_fma(eps, m, xp[e]); xp[e] = go ? v : xp[e]; });
                dense_stream<NE, HASMU>(Pfrag, cx.lane, xp, ga, ldsMu);              // ga = +P (x - mu)
                const double nkf = l + 1 < nl ? -eps : -halfe;
                mom_update<NE>(momw, [&](int e, double m) { const double v = kd_fma(nkf, (double)ga[e >> 2][e & 3], m); return go ? v : m; });
            }
#pragma unroll
            for (int tt = 0; tt < MT; ++tt) ga[tt] = -ga[tt];                        // the proposal's gradient, -P (x' - mu)
            double red[2], l1 = 0.0, k1 = 0.0;
            mom_read<NE>(momw, [&](int e, double m) {
                const double d = HASMU ? xp[e] - ldsMu[4 * e + cx.q] : xp[e];
                l1 = l1 + d * (double)ga[e >> 2][e & 3];                             // lt' = c + 1/2 (x'-mu).g'   (HMC.jl:157)
                k1 = k1 + m * m;
            });
            red[0] = l1; red[1] = k1;
            mreduce<2>(red, cx.lane);
            ltp = p.gconst + 0.5 * red[0];
            const double H1 = ltp - 0.5 * red[1];                                    // HMC.jl:159
            const double ratio = H1 - H0;                                            // HMC.jl:161
            const double ex = kd_exp(ratio);
            const double a = 1.0 < ex ? 1.0 : ex;                                    // HMC.jl:163
            const double u = kd_accept_uniform(kd_stream_block(p.seed, gchain, t, (uint32_t)((p.D + 1) >> 1)));
            acc = u < a;                                                             // HMC.jl:165
            if (da) da_update(p, tn, (long long)t + 1, a);                           // HMC.jl:225-249

        } else if constexpr (SAMPLER == KLARA_SAMPLER_MALA) {
            // iterate/MALA.jl:78-128.  The proposal overwrites the value registers as its normals are drawn; the current gradient (accumulators) is
            // consumed by the same pass; the current value waits in the lane's LDS column for the backward term: no vector beyond x and P x is held in registers.
            const double h = tn.step, halfh = 0.5 * h, sq = __builtin_sqrt(h), half_inv_h = 0.5 * (1.0 / h);
            double s1 = 0.0;
            // the normals go through the lane's LDS column like HMC's momentum: drawn first, consumed in groups of 8 — the transform's ~40 live
            // registers and the pass over value + gradient do not overlap (drawn straight into the pass, NE = 64 comes out with 530 B of scratch and
            // 18 % slower: measured again in round 5)
            mnormals_lds<NE>(cx, p.seed, gchain, t, momw);
            // Round 5: nothing of this pass comes from memory any more.  The CURRENT gradient is in the accumulators (it stays there between transitions
            // like HMC's; round 4 re-read it from GR here, 8 loads in flight per group: eight exposed memory round trips per transition at one wavefront per
            // SIMD), and the current value takes the place of each normal in the lane's LDS column as it is consumed — the backward term below reads it
            // from there instead of re-reading X (another eight round trips), and so does a lane whose proposal is rejected.
            {
#pragma unroll
                for (int e0 = 0; e0 < NE; e0 += 8) {
                    double zz[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) zz[j] = momw[(e0 + j) * 64];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int e = e0 + j;
                        momw[e * 64] = xp[e];                                         // (the lane's own column: DS operations of a lane stay in order)
                        const double mu = xp[e] + halfh * (double)ga[e >> 2][e & 3];  // MALA.jl:83
                        xp[e] = mu + sq * zz[j];                                      // MALA.jl:84
                        const double q1 = mu - xp[e];
                        s1 = s1 + (q1 * q1) * half_inv_h;                             // MALA.jl:90
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            KLARA_BIG_PHASE();
            dense_stream<NE, HASMU>(Pfrag, cx.lane, xp, ga, ldsMu);                  // MALA.jl:86
            KLARA_BIG_PHASE();
#pragma unroll
            for (int tt = 0; tt < MT; ++tt) ga[tt] = -ga[tt];
            double l1 = 0.0, s2 = 0.0, red[3];
            {
#pragma unroll
                for (int e0 = 0; e0 < NE; e0 += KLARA_BIG_XC_CHUNK) {      // the current value from the lane's LDS column, 8 elements at a time
                    double xc[KLARA_BIG_XC_CHUNK];
#pragma unroll
                    for (int j = 0; j < KLARA_BIG_XC_CHUNK; ++j) xc[j] = momw[(e0 + j) * 64];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < KLARA_BIG_XC_CHUNK; ++j) {
                        const int e = e0 + j;
                        const double d = HASMU ? xp[e] - ldsMu[4 * e + cx.q] : xp[e];
                        const double gpe = (double)ga[e >> 2][e & 3];
                        l1 = l1 + d * gpe;
                        const double mup = xp[e] + halfh * gpe;                       // MALA.jl:91
                        const double q2 = mup - xc[j];
                        s2 = s2 + (q2 * q2) * half_inv_h;                             // MALA.jl:92
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            red[0] = l1; red[1] = s1; red[2] = s2;
            KLARA_BIG_PHASE();
            mreduce<3>(red, cx.lane);
            ltp = p.gconst + 0.5 * red[0];
            double ratio = ltp - lt;                                                  // MALA.jl:88
            ratio += red[1];
            ratio -= red[2];
            acc = ratio > 0.0;                                                        // MALA.jl:94
            if (__any(!acc && ratio > KD_LOG_UMIN_GUARD)) {      // (wave-uniform: see the note at the commit below; at or below the guard no uniform accepts)
                const double u = kd_accept_uniform(kd_stream_block(p.seed, gchain, t, (uint32_t)((p.D + 1) >> 1)));
                acc = acc || ratio > kd_log_u01(u);
            }
        } else if constexpr (SLICE) {
            // iterate/SliceSampler.jl:60-109 — the 16 chains of the tile out of lockstep, one pass over P per probe (slice_dense_free, klara_dense.h)
            slice_dense_free<NE>(p, cx, gchain, t, xp, ltp, stuck, [&](const double (&x)[NE]) {
                dense_stream<NE, HASMU>(Pfrag, cx.lane, x, ga, ldsMu);
                double l1 = 0.0, r1[1];
#pragma unroll
                for (int e = 0; e < NE; ++e) l1 = l1 + (HASMU ? x[e] - ldsMu[4 * e + cx.q] : x[e]) * (-(double)ga[e >> 2][e & 3]);
                r1[0] = l1;
                mreduce<1>(r1, cx.lane);
                return p.gconst + 0.5 * r1[0];
            });
            acc = true;                                  // the slice sampler always moves (SliceSampler.jl:108)
        } else {
            // iterate/MH.jl:72-124
            mnormals_each<NE, KLARA_BIG_NORMALS_WAYS_MH>(cx, p.seed, gchain, t, [&](int e, double z) {
                const double sg = ldsSig[4 * e + cx.q];                               // (0 past D)
                momw[e * 64] = xp[e];                                                 // the current value: what a rejecting lane goes back to
                xp[e] = xp[e] + sg * z;                                               // MH.jl:79
            });
            dense_stream<NE, HASMU>(Pfrag, cx.lane, xp, ga, ldsMu);                  // MH.jl:81
            double l1 = 0.0, red[1];
#pragma unroll
            for (int e = 0; e < NE; ++e) l1 = l1 + (HASMU ? xp[e] - ldsMu[4 * e + cx.q] : xp[e]) * (-(double)ga[e >> 2][e & 3]);
            red[0] = l1;
            mreduce<1>(red, cx.lane);
            ltp = p.gconst + 0.5 * red[0];
            const double ratio = ltp - lt;                                            // MH.jl:83
            acc = ratio > 0.0;                                                        // MH.jl:97
            if (__any(!acc && ratio > KD_LOG_UMIN_GUARD)) {      // (wave-uniform: see the note at the commit below; at or below the guard no uniform accepts)
                const double u = kd_accept_uniform(kd_stream_block(p.seed, gchain, t, (uint32_t)((p.D + 1) >> 1)));
                acc = acc || ratio > kd_log_u01(u);
            }
        }

        // Fold, commit, reload: NO divergent branch around the value / gradient arrays.  With ~500 live registers the allocator parks part of them in
        // accumulator registers, and a copy it places at the head of a join block runs under the branch's execution mask, before the mask is restored:
        // k_dense_big<MH, 48, mean> lost element 15 of every chain that had just rejected that way (ROCm 7.2; found by the randomised parity jobs when the
        // normals changed the allocation).  So every lane takes every step under a wave-uniform condition; a lane with nothing to do addresses out of
        // bounds (loads return 0, stores are dropped) and keeps its registers through selects.
        if (do_sum && __any(acc && held > 0)) {          // leaving a state after `held` saved steps: fold it into the sums (the OLD value is in X, or in the LDS column)
            const bool fold = acc && held > 0;
            const double hf = (double)held;
            const __amdgpu_buffer_rsrc_t ws = mwin<NE>(cx, p.sum, 0, p.D), wq = mwin<NE>(cx, p.sumsq, 0, p.D);
            const int nv = cx.nv_here();
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const unsigned o = fold ? cx.off(e, nv) : KLARA_BUF_OOB;
                const double xo = XONCE ? momw[e * 64] : __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(wX, o, 0, 0));     // (the lane's LDS column still holds the value the chain is leaving)
                const double sv = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(ws, o, 0, 0));
                const double qv = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(wq, o, 0, 0));
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(kd_uint2, sv + hf * xo), ws, o, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(kd_uint2, qv + hf * (xo * xo)), wq, o, 0, 0);
            }
            held = fold ? 0 : held;
        }
        KLARA_BIG_PHASE();
        if ((!XONCE || NEEDG) && __any(acc)) {           // commit (HMC.jl:166-176): what a later reject re-reads
            const int nv = cx.nv_here();
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const unsigned o = acc ? cx.off(e, nv) : KLARA_BUF_OOB;
                constexpr int AUX = SAMPLER == KLARA_SAMPLER_HMC ? KLARA_BIG_HMC_AUX : (SAMPLER == KLARA_SAMPLER_MALA ? KLARA_BIG_G_AUX : 0);
                if (!XONCE) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(kd_uint2, xp[e]), wX, o, 0, AUX);
                if (NEEDG) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(kd_uint2, (double)ga[e >> 2][e & 3]), wG, o, 0, AUX);
            }
        }
        lt = acc ? ltp : lt;
        KLARA_BIG_PHASE();
        if (__any(!acc)) {                               // the registers of a lane that rejected hold the proposal: back to the committed state
            const int nv = cx.nv_here();
            constexpr int RC = KLARA_BIG_RELOAD_CHUNK;
#pragma unroll
            for (int e0 = 0; e0 < NE; e0 += RC) {        // RC elements at a time: the loads of a group in flight, then its selects (not 2 NE loaded values live at once)
                double xc[RC], gc[RC];
#pragma unroll
                for (int j = 0; j < RC; ++j) {
                    const unsigned o = acc ? KLARA_BUF_OOB : cx.off(e0 + j, nv);
                    if (XLDS) xc[j] = momw[(e0 + j) * 64];                            // (MALA: the current value is still in the lane's LDS column)
                    else xc[j] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(wX, o, 0, 0));
                    if (KEEPG) gc[j] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(wG, o, 0, 0));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < RC; ++j) {
                    const int e = e0 + j;
                    xp[e] = acc ? xp[e] : xc[j];
                    if (KEEPG) ga[e >> 2][e & 3] = acc ? ga[e >> 2][e & 3] : gc[j];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        nacc += acc ? 1ull : 0ull;
        tn.accepted += (p.cnt && acc && !SLICE) ? 1 : 0;          // (the slice sampler never counts accepts)
        if (accept_out != nullptr) {                     // (one byte per chain from its q = 0 lane; the other lanes address out of bounds)
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
                    if constexpr (KEEPG) {
#pragma unroll
                        for (int e = 0; e < NE; ++e)
                            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(kd_uint2, (double)ga[e >> 2][e & 3]), wh, cx.off(e, nv), 0, 0);
                    } else {                             // (MALA: the committed gradient is in GR — the accumulators may hold a rejected proposal's)
#pragma unroll
                        for (int e0 = 0; e0 < NE; e0 += 8) {
                            double gc[8];
#pragma unroll
                            for (int j = 0; j < 8; ++j) gc[j] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(wG, cx.off(e0 + j, nv), 0, 0));
#pragma unroll
                            for (int j = 0; j < 8; ++j) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(kd_uint2, gc[j]), wh, cx.off(e0 + j, nv), 0, 0);
                        }
                    }
                }
                if (p.hist_lt != nullptr) {
                    const __amdgpu_buffer_rsrc_t wl = mwin<NE>(cx, p.hist_lt, col * p.nchains, 1);                // (row `col`, the tile's chains)
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(kd_uint2, lt), wl,
                                                          (cx.chain_ok && cx.q == 0) ? (unsigned)cx.cl * 8u : KLARA_BUF_OOB, 0, 0);
                }
            }
        }
    }

    if (SLICE && stuck && cx.chain_ok && cx.q == 0) klara_raise(p.error_flag, KLARA_ERR_SLICE_STUCK);
    if (XONCE) {                                         // the committed value, once per launch
        const int nv = cx.nv_here();
#pragma unroll
        for (int e = 0; e < NE; ++e) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(kd_uint2, xp[e]), wX, cx.off(e, nv), 0, 0);
    }
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

// initialize! for the streamed layouts: g = -P (x - mu), lt = c + 1/2 (x - mu).g, finiteness asserts
template <int NE, bool HASMU = false>
__global__ __launch_bounds__(256) void k_dense_init_big(const KParams p, const double* __restrict__ Pfrag, int needgrad)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MT = NE / 4;
    double* const ldsMuW = reinterpret_cast<double*>(smem);
    if (HASMU) { for (int i = threadIdx.x; i < 4 * NE; i += blockDim.x) ldsMuW[i] = Pfrag[(size_t)MT * NE * 64 + i]; }
    const double* const ldsMu = ldsMuW;
    __syncthreads();
    const MfmaCtx<NE> cx = make_mctx<NE>(p);
    double x[NE], red[1];
    kd_double4 ga[MT];
    mload<NE>(cx, p.X, p.D, x);
    dense_stream<NE, HASMU>(Pfrag, cx.lane, x, ga, ldsMu);
    double l1 = 0.0;
    bool bad = false;
    const __amdgpu_buffer_rsrc_t wG = mwin<NE>(cx, p.GR, 0, p.D);
    const int nv = cx.nv_here();
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const double g = -ga[e >> 2][e & 3];
        l1 = l1 + (HASMU ? x[e] - ldsMu[4 * e + cx.q] : x[e]) * g;
        if (needgrad) {
            bad = bad || !kfinite(g);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(kd_uint2, g), wG, cx.off(e, nv), 0, 0);
        }
    }
    red[0] = l1;
    mreduce<1>(red, cx.lane);
    const double lt = p.gconst + 0.5 * red[0];
    bad = bad || (cx.chain_ok && !kfinite(lt));
    if (cx.chain_ok && cx.q == 0) p.LT[cx.chain] = lt;
    if (bad) klara_raise(p.error_flag, KLARA_ERR_NONFINITE_INIT);
}
