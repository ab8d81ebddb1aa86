// klara_dense.h — dense-Gaussian target on the FP64 matrix cores (layout kind 1, "MFMA-transposed").
//
// Target (builder-defined; Klara ships no dense example — SURVEY F8, §8(d) cfg 3):
//     lt(x) = c - 1/2 (x-mu)' P (x-mu),   grad(x) = -P (x-mu),   P = D x D precision matrix; HASMU = false is the mu = 0 form
//     (no subtraction anywhere), HASMU = true reads mu from LDS where the difference is formed.
//
// One wavefront carries 16 chains.  Lane l = (q = l >> 4, cl = l & 15) belongs to chain cl of the tile
// and holds the NE = ceil(D/4) elements { 4e + q : e = 0..NE-1 } of every per-chain vector.  The
// gradient of all 16 chains is one transposed GEMM  G' = P * X'  (M = D rows of P, N = 16 chains,
// K = D) on v_mfma_f64_16x16x4_f64:
//     A operand (one f64/lane) = P[16t + (l&15)][4kk + (l>>4)]        from LDS, fragment-ordered
//     B operand (one f64/lane) = x[chain l&15][4kk + (l>>4)]           = the lane's own element kk
//     D (4 f64/lane), tile t, reg r: row (l>>4) + 4r, col l&15         = element 4t+r of the lane
// so the accumulator registers ARE the lane's gradient elements in the same distribution the next
// leapfrog needs as its B operand: no transpose, no LDS round trip for the state.  K = D exactly
// (no padding on K for D % 4 == 0); M is padded to 16*ceil(D/16) rows of zeros.
// Accumulation order per output element = one fma chain over k ascending (checked on hardware against
// the oracle's chain, tests/test_gpu_parity.py::test_mfma_f64_order).
//
// Registers: the kernel keeps only the proposal (xp, gp=accumulators, momentum) live; the current
// state stays in HBM/L2 and is re-read on reject, which keeps NE = 25 under 256 VGPRs (2 waves/SIMD).
#pragma once
#include "klara_kernels.h"

typedef double kd_double4 __attribute__((ext_vector_type(4)));

template <int NE>
struct MfmaCtx {
    static constexpr int MT = (NE + 3) / 4;
    int lane, q, cl;
    long long chain, first_chain;      // first_chain: chain 0 of the wavefront's tile (wave-uniform)
    int here;                          // chains of the tile that exist
    bool chain_ok;
    int nv;          // elements e < nv of this lane are real: 4 e + q < D on an existing chain (one register instead of NE lane masks)
    unsigned voff0;  // byte offset of element 0 of this lane inside the tile's window
    __device__ __forceinline__ bool valid(int e) const { return e < nv; }
    // Padding is zeroed where it enters (loads return 0 outside the window, the normals of missing elements are set to 0, the padded
    // rows of P are 0): every missing element of every per-chain vector then stays +-0 through the transition, its terms add +-0 to
    // the sums, and no sum needs a validity select.  Only the byte offsets of loads and stores depend on nv.
    // (nv_ comes from nv_here() at the access group's site; the base offset is hidden from the optimiser the same way — otherwise voff0 + 32 e
    // is hoisted out of the transition loop for every e: NE registers that the MFMA loop then pushes into scratch)
    __device__ __forceinline__ unsigned off_fresh(int e, int nv_) const
    {
        unsigned vb = voff0;
        __asm__ volatile("" : "+v"(vb));
        return e < nv_ ? vb + 32u * (unsigned)e : KLARA_BUF_OOB;
    }
    // (the streamed kernels of klara_dense_big.h — one wavefront per SIMD, 512 registers — keep the plain form: the extra instruction per access
    // cost their MH instantiation 3 %)
    __device__ __forceinline__ unsigned off(int e, int nv_) const { return e < nv_ ? voff0 + 32u * (unsigned)e : KLARA_BUF_OOB; }
    // nv behind an empty asm: the NE offsets of an access group are formed where they are used (2 instructions each) instead of
    // being hoisted out of the transition loop as NE loop-invariant registers that the MFMA loop then forces into scratch
    __device__ __forceinline__ int nv_here() const { int n = nv; __asm__ volatile("" : "+v"(n)); return n; }
    // the lane's chain index, re-formed where it is used (a scalar base + the lane's column) instead of living in two registers through the MFMA loops
    __device__ __forceinline__ long long chain_here() const { int c = cl; __asm__ volatile("" : "+v"(c)); return first_chain + c; }
};

template <int NE>
__device__ __forceinline__ MfmaCtx<NE> make_mctx(const KParams& p)
{
    MfmaCtx<NE> c;
    c.lane = threadIdx.x & 63;
    c.q = c.lane >> 4;
    c.cl = c.lane & 15;
    const long long wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    c.first_chain = wave * 16;
    const long long left = p.nchains - c.first_chain;
    c.here = left < 16 ? (left > 0 ? (int)left : 0) : 16;
    c.chain = c.first_chain + c.cl;
    c.chain_ok = c.cl < c.here;
    c.nv = c.chain_ok ? (p.D - c.q + 3) / 4 : 0;
    if (c.nv > NE) c.nv = NE;
    c.voff0 = (unsigned)((c.cl * p.D + c.q) * 8);
    return c;
}

// the tile's rows of a (chains x D) array as one buffer window (scalar descriptor; accesses outside it return 0 / are dropped)
template <int NE>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t mwin(const MfmaCtx<NE>& c, const gdouble* base, long long row0, int D)
{
    // (every term is wave-uniform; the explicit readfirstlane keeps the descriptor in scalar registers where the compiler cannot
    // prove it — a descriptor in vector registers costs a readfirstlane loop around every access)
    const unsigned long long a = (unsigned long long)(base + (row0 + c.first_chain) * D);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0,
                                             __builtin_amdgcn_readfirstlane(c.here * D * 8), 0x00020000);
}
template <int NE>
__device__ __forceinline__ void mload(const MfmaCtx<NE>& c, const gdouble* base, int D, double (&v)[NE], long long row0 = 0)
{
    const __amdgpu_buffer_rsrc_t w = mwin<NE>(c, base, row0, D);
    const int nv = c.nv_here();
#pragma unroll
    for (int e = 0; e < NE; ++e) v[e] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(w, c.off_fresh(e, nv), 0, 0));
}
#ifndef KLARA_DENSE_COMMIT_AUX
#define KLARA_DENSE_COMMIT_AUX 0      // cache-policy bits of the X / GR stores of an accepted transition (2 = nt)
#endif
template <int NE, int AUX = 0>
__device__ __forceinline__ void mstore(const MfmaCtx<NE>& c, gdouble* base, int D, const double (&v)[NE], long long row0 = 0)
{
    const __amdgpu_buffer_rsrc_t w = mwin<NE>(c, base, row0, D);
    const int nv = c.nv_here();
#pragma unroll
    for (int e = 0; e < NE; ++e) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(kd_uint2, v[e]), w, c.off_fresh(e, nv), 0, AUX);
}

// all-reduce over the 4 lanes (q = 0..3) of a chain: xor 16 then xor 32 — the canonical tree
// (q0+q1)+(q2+q3) of the oracle's kind-1 layout.
template <int N>
__device__ __forceinline__ void mreduce(double (&v)[N], int lane)
{
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = v[i] + bperm_xor(v[i], lane, 16);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = v[i] + bperm_xor(v[i], lane, 32);
}

// normals for the lane's elements: element e <-> dim i = 4e+q <-> pair index i>>1 = 2e + (q>>1), branch q&1 (cos for even dims, sin for odd
// dims); the 64 bits of pair index p are half (p >> 3) & 1 of block slot (p & 7) + 8 (p >> 4) (detmath.h kd_normal_pair_at).
// Lanes q and q^1 (lane ^ 16) of a chain need the two halves of the SAME Box-Muller pairs.  Instead of both lanes evaluating every pair,
// elements are taken two at a time: the even-q lane evaluates the pair of element e, the odd-q lane the pair of element e+1, and they swap the
// halves the partner needs with one ds_bpermute — half the transforms, identical values.  In loop slot e (even) the lane's pair index is
// 2 (e + odd) + sh: its block slot is a compile-time number + (2 odd + sh) and its half is (e >> 2) & 1 for both lanes, so the pair 8 further on
// (loop slot e + 4, same lane) takes the words (z, w) the block of slot e left behind: one Philox block per two slots.  (Pairs past ceil(D/2)
// are layout padding: whatever they draw is discarded; the accept draw is taken explicitly by these kernels.)
struct MPairStash { uint32_t w[4]; };
__device__ __forceinline__ void mpair_normals(unsigned long long seed, unsigned long long gchain, unsigned long long t, int e, uint32_t lane_slot,
                                              MPairStash& st, double& z0, double& z1)
{
    const int si = (e >> 1) & 1;
    uint32_t wa, wb;
    if ((e >> 2) & 1) { wa = st.w[2 * si]; wb = st.w[2 * si + 1]; }
    else {
        const uint32_t base = ((2u * (uint32_t)e) & 7u) | (((2u * (uint32_t)e) >> 4) << 3);
        const kd_u32x4 b = kd_stream_block(seed, gchain, t, base + lane_slot);
        wa = b.x; wb = b.y; st.w[2 * si] = b.z; st.w[2 * si + 1] = b.w;
    }
    double u1, lg;
    kd_normal_pair_w(wa, wb, &z0, &z1, &u1, &lg);
}
// N Box-Muller evaluations with their statements interleaved — the same operations in the same order within each (kd_normal_pair_w, detmath.h:
// kd_u44, kd_log_u01, kd_sqrt_radicand, kd_sincos2pi_bits written out statement by statement over an index j), so bit-identical to N separate calls.
// A wavefront that is alone on its SIMD (klara_dense_big.h) has nothing else to fill the latencies of one evaluation's dependent chains and table
// lookups, and the compiler does not interleave separate calls by itself (it schedules for register pressure there).
template <int N>
__device__ __forceinline__ void mpair_normals_n(const uint32_t (&wa)[N], const uint32_t (&wb)[N], double (&z0)[N], double (&z1)[N])
{
#define KD_EACH _Pragma("unroll") for (int j = 0; j < N; ++j)
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double B0 = 0x1.5555555555555p-2, B1 = -0.25, B2 = 0x1.999999999999ap-3, B3 = -0x1.5555555555555p-3,
                 B4 = 0x1.2492492492492p-3, B5 = -0.125;
    const double S1 = -0x1.5555555555555p-3, S2 = 0x1.1111111111111p-7, S3 = -0x1.a01a01a01a01ap-13;
    const double C2 = 0x1.5555555555555p-5, C3 = -0x1.6c16c16c16c17p-10;
    double u1[N], z[N], invc[N], logc[N], Ct[N], St[N], uu[N], cc[N];
    uint32_t i[N], jt[N]; int k[N];
    // kd_u44, kd_log_u01_reduce, the angle's bits (kd_angle_bits20) and both table lookups first: their latencies run under everything below
    KD_EACH u1[j] = kd_u44(wa[j], wb[j]);
    KD_EACH kd_log_u01_reduce(u1[j], &i[j], &k[j], &z[j]);
    KD_EACH { invc[j] = KD_LOGTAB(2 * i[j]); logc[j] = KD_LOGTAB(2 * i[j] + 1); }
    KD_EACH {
        const uint64_t ub = kd_angle_bits20(wb[j]);
        const uint32_t uh = (uint32_t)(ub >> 32);
        jt[j] = (uh >> 12) & 255u;
        uu[j] = kd_u2d(ub);
        cc[j] = kd_u2d((uint64_t)((uh & 0xfffff000u) | 0x00000800u) << 32);
    }
    KD_EACH { Ct[j] = KD_SCTAB(2 * jt[j]); St[j] = KD_SCTAB(2 * jt[j] + 1); }
    // kd_log_u01_finish
    double r[N], dk[N], w[N], hi[N], lo[N], r2[N], q[N], lg[N];
    KD_EACH r[j] = kd_fma(z[j], invc[j], -1.0);
    KD_EACH dk[j] = (double)k[j];
    KD_EACH w[j] = kd_fma(dk[j], ln2_hi, logc[j]);
    KD_EACH hi[j] = w[j] + r[j];
    KD_EACH lo[j] = kd_fma(dk[j], ln2_lo, (w[j] - hi[j]) + r[j]);
    KD_EACH r2[j] = r[j] * r[j];
    KD_EACH q[j] = kd_fma(r2[j], kd_fma(r2[j], kd_fma(r[j], B5, B4), kd_fma(r[j], B3, B2)), kd_fma(r[j], B1, B0));
    KD_EACH lg[j] = hi[j] + kd_fma(r[j] * r2[j], q[j], kd_fma(r2[j], -0.5, lo[j]));
    // kd_sincos2pi_bits up to the table terms (independent of the logarithm: fills the rsq's latency below)
    double t[N], y[N], zz[N], sy[N], dc[N];
    KD_EACH t[j] = uu[j] - cc[j];
    KD_EACH y[j] = kd_fma(t[j], KD_TWOPI_HI, kd_fma(t[j], KD_TWOPI_LO, KD_TWOPI_2M53));
    // kd_sqrt_radicand(-2 lg)
    double ya[N], rs[N], g[N], h[N], e[N], d[N];
    KD_EACH ya[j] = -2.0 * lg[j];
    KD_EACH rs[j] = __builtin_amdgcn_rsq(ya[j]);
    KD_EACH zz[j] = y[j] * y[j];
    KD_EACH sy[j] = kd_fma(y[j] * zz[j], kd_fma(zz[j], kd_fma(zz[j], S3, S2), S1), y[j]);
    KD_EACH dc[j] = zz[j] * kd_fma(zz[j], kd_fma(zz[j], C3, C2), -0.5);
    KD_EACH { g[j] = ya[j] * rs[j]; h[j] = 0.5 * rs[j]; }
    KD_EACH e[j] = kd_fma(-h[j], g[j], 0.5);
    KD_EACH { g[j] = kd_fma(g[j], e[j], g[j]); h[j] = kd_fma(h[j], e[j], h[j]); }
    KD_EACH d[j] = kd_fma(-g[j], g[j], ya[j]);
    KD_EACH g[j] = kd_fma(d[j], h[j], g[j]);
    KD_EACH d[j] = kd_fma(-g[j], g[j], ya[j]);
    KD_EACH g[j] = kd_fma(d[j], h[j], g[j]);                                  // rad
    KD_EACH {
        const double cs = kd_fma(-St[j], sy[j], kd_fma(Ct[j], dc[j], Ct[j]));
        const double sn = kd_fma(Ct[j], sy[j], kd_fma(St[j], dc[j], St[j]));
        z0[j] = g[j] * cs;
        z1[j] = g[j] * sn;
    }
#undef KD_EACH
}

template <int NE>
__device__ __forceinline__ void mnormals(const MfmaCtx<NE>& c, unsigned long long seed,
                                         unsigned long long gchain, unsigned long long t,
                                         double (&z)[NE])
{
    const uint32_t sh = (uint32_t)(c.q >> 1);
    const bool odd = (c.q & 1) != 0;
    const int nv = c.nv_here();
    MPairStash st = { { 0u, 0u, 0u, 0u } };
#pragma unroll
    for (int e = 0; e + 1 < NE; e += 2) {
        double z0, z1;
        mpair_normals(seed, gchain, t, e, (odd ? 2u : 0u) + sh, st, z0, z1);
        const double recv = bperm_xor(odd ? z0 : z1, c.lane, 16);
        z[e] = e < nv ? (odd ? recv : z0) : 0.0;              // even q: cos half of pair(e);   odd q: sin half of pair(e) from the partner
        z[e + 1] = e + 1 < nv ? (odd ? z1 : recv) : 0.0;  // even q: cos half of pair(e+1) from the partner; odd q: sin half of pair(e+1)
        __builtin_amdgcn_sched_barrier(0);   // keep the unrolled Philox/Box-Muller bodies from interleaving (VGPR pressure)
    }
    if (NE & 1) {                            // the last element alone: both lanes of a pair evaluate pair index 2 (NE - 1) + sh
        constexpr uint32_t p0 = 2u * (uint32_t)(NE - 1);
        const kd_u32x4 b = kd_stream_block(seed, gchain, t, ((p0 & 7u) | ((p0 >> 4) << 3)) + sh);
        constexpr bool hb = ((p0 >> 3) & 1u) != 0u;
        double z0, z1, u1, lg;
        kd_normal_pair_w(hb ? b.z : b.x, hb ? b.w : b.y, &z0, &z1, &u1, &lg);
        z[NE - 1] = NE - 1 < nv ? (odd ? z1 : z0) : 0.0;
    }
}

// g = -(P x) for the wave's 16 chains.  ldsP: fragment-ordered P, (MT*NE) fragments of 64 doubles.
//
// Tail tile.  When NE % 4 == 1 (D = 97..100 -> NE = 25) the last 16-row tile would hold only 4 real rows — one element
// per lane — and 12 rows of zeros.  Those 4 rows go through v_mfma_f64_4x4x4_4b instead (4 blocks of 4x4x4, 16 cycles
// instead of 64): on gfx950 its operands sit on lane 16k + 4b + i (A_b[i][k]), 16k + 4b + j (B_b[k][j]) and its result on
// lane 16i + 4b + j (D_b[i][j]) (scripts/probe_mfma4.hip), so with block b = chains 4b..4b+3 the B operand is again the
// lane's own element kk, and the result D_b[i][j] = g_{16*MTF + i} of chain 4b + j lands on lane (q = i, chain) — the
// lane that owns that element.  No shuffle, no extra registers (one f64 accumulator instead of four), the same LDS
// footprint (the tail's A fragments take the place of the padded tile's), 10.7 % less matrix-pipe time at D = 100.
// Its accumulation order is the same k-ascending fma chain (tests/test_gpu_parity.py::test_mfma_f64_4x4x4_order).
// HASMU: the B operand is x - mu, mu read per k-step from LDS (ldsMu[4 kk + q], zero for missing elements).
// NEG = false returns +P (x - mu) (the leapfrog folds the sign into its fma: p = fma(-k, P x, p)).
template <int NE, bool HASMU = false, bool NEG = true>
__device__ __forceinline__ void dense_grad(const double* __restrict__ ldsP, int lane,
                                           const double (&x)[NE], double (&g)[4 * ((NE + 3) / 4)], const double* ldsMu = nullptr)
{
    constexpr int MT = (NE + 3) / 4;
    constexpr bool TAIL = (NE % 4) == 1;
    constexpr int MTF = TAIL ? MT - 1 : MT;         // tiles on v_mfma_f64_16x16x4
    kd_double4 acc[MTF];
    double acc_t = 0.0;
#pragma unroll
    for (int t = 0; t < MTF; ++t) acc[t] = (kd_double4){ 0.0, 0.0, 0.0, 0.0 };
    // software pipeline: the A fragments of k-step kk+1 are fetched from LDS while the MT MFMAs of
    // k-step kk issue; sched_barrier keeps hipcc from hoisting all NE*MT LDS reads (VGPR blow-up).
    double a_cur[MT], a_nxt[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) a_cur[t] = ldsP[(t * NE) * 64 + lane];
#pragma unroll
    for (int kk = 0; kk < NE; ++kk) {
        const double b = HASMU ? x[kk] - ldsMu[4 * kk + (lane >> 4)] : x[kk];
        if (kk + 1 < NE) {
#pragma unroll
            for (int t = 0; t < MT; ++t) a_nxt[t] = ldsP[(t * NE + kk + 1) * 64 + lane];
        }
#pragma unroll
        for (int t = 0; t < MTF; ++t)
            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_cur[t], b, acc[t], 0, 0, 0);
        if (TAIL) acc_t = __builtin_amdgcn_mfma_f64_4x4x4f64(a_cur[MT - 1], b, acc_t, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < MT; ++t) a_cur[t] = a_nxt[t];
    }
#pragma unroll
    for (int t = 0; t < MTF; ++t) {
        g[4 * t + 0] = NEG ? -acc[t][0] : acc[t][0]; g[4 * t + 1] = NEG ? -acc[t][1] : acc[t][1];
        g[4 * t + 2] = NEG ? -acc[t][2] : acc[t][2]; g[4 * t + 3] = NEG ? -acc[t][3] : acc[t][3];
    }
    if (TAIL) { g[4 * MTF + 0] = NEG ? -acc_t : acc_t; g[4 * MTF + 1] = 0.0; g[4 * MTF + 2] = 0.0; g[4 * MTF + 3] = 0.0; }
}

// The slice sampler on a dense target, the chains of a tile out of lockstep (slice_free_machine, klara_kernels.h): a probe is one matrix pass over the
// 16 chains of the tile, lt = c + 1/2 (x - mu).g, and each chain takes from it the probe its own coordinate and stage ask for (round 4: one stage of one
// coordinate per pass, waiting for the slowest of the 16, at five call sites of the pass).  Coordinate i = 4 e + q lives on lane q = i & 3 of its chain as
// register e = i >> 2.  probe(x) returns the log-target of the lane's chain at x, the same in the chain's 4 lanes.
template <int NE, class Probe>
__device__ __forceinline__ void slice_dense_free(const KParams& p, const MfmaCtx<NE>& cx, unsigned long long gchain, unsigned long long t,
                                                 double (&xp)[NE], double& cur, bool& stuck, Probe probe)
{
    slice_free_machine(p, cx.chain_ok, gchain, t, cur, stuck,
        [&](int i, double& xs, double& ws) {
            const int qo = i & 3, eo = i >> 2;
            double xi_l = 0.0;
#pragma unroll
            for (int e = 0; e < NE; ++e) xi_l = (e == eo) ? xp[e] : xi_l;
            xs = lane_bcast(xi_l, cx.cl + 16 * qo);
            ws = p.vecparam[i];
        },
        [&](int i, bool on, double cand) {
            const int qo = i & 3, eo = i >> 2;
            const bool owner = on && cx.q == qo;
#pragma unroll
            for (int e = 0; e < NE; ++e) xp[e] = (owner && e == eo) ? cand : xp[e];
        },
        [&]() { return probe(xp); });
}

// PLAIN: nothing counts proposals or tunes (VanillaMCTuner, not verbose — BASELINE cfg 3): the step is the job's scalar step0, no tuner
// state is loaded, carried through the launch or written back (13 registers per lane that the generic instantiation spills: 80 B of scratch)
template <int SAMPLER, int NE, bool DA, bool HASMU = false, bool PLAIN = false>
__global__ __launch_bounds__(512)
void k_dense_transitions(const KParams* __restrict__ pp, const KLaunch kl, const double* __restrict__ Pfrag)
{
    const KParams& p = *pp;
    guchar* const accept_out = p.accept != nullptr ? p.accept + kl.t0 * (unsigned long long)p.nchains : nullptr;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MT = (NE + 3) / 4;
    constexpr int NG = 4 * MT;
    double* ldsP = reinterpret_cast<double*>(smem);
    for (int i = threadIdx.x; i < MT * NE * 64; i += blockDim.x) ldsP[i] = Pfrag[i];
    const double* const ldsMu = ldsP + MT * NE * 64;             // mu[4 e + q] at [4 e + q], zero-padded (HASMU only)
    if (HASMU) { for (int i = threadIdx.x; i < 4 * NE; i += blockDim.x) ldsP[MT * NE * 64 + i] = Pfrag[MT * NE * 64 + i]; }
    kd_tables_to_lds();          // (also the barrier for ldsP)

    const MfmaCtx<NE> cx = make_mctx<NE>(p);
    // x - mu of the lane's element e (lt = c + 1/2 (x-mu).g); missing elements: 0 - 0
    const auto dx = [&](const double (&v)[NE], int e) { return HASMU ? v[e] - ldsMu[4 * e + cx.q] : v[e]; };
    const auto gch = [&]() { return (unsigned long long)(p.chain_offset + cx.chain_here()); };       // global chain id: the Philox subsequence
    const long long tix = p.pooled ? 0 : (cx.chain_ok ? cx.chain : 0);
    constexpr bool da = DA;   // dual averaging is a separate instantiation: the masked leapfrog loop costs registers
    static_assert(!(PLAIN && DA), "dual averaging tunes");
    const bool cnt = !PLAIN && p.cnt != 0;
    TuneRegs tn = { p.step0, 0, 0, 0, 0, 0.0, 0.0 };
    if (!PLAIN) tn = { p.tune_step[tix], p.tune_accepted[tix], p.tune_proposed[tix], p.tune_totproposed[tix], 0, 0.0, 0.0 };
    if (da) { tn.epsbar = p.da_epsbar[tix]; tn.hbar = p.da_hbar[tix]; }
    tn.phase = cnt ? (int)(tn.proposed % p.period) : 0;
    int sphase = kl.save_phase0;
    long long scol = kl.save_col0;
    double lt = cx.chain_ok ? p.LT[cx.chain] : 0.0;
    unsigned long long nacc = 0;
    bool stuck = false;                                  // slice sampler: step-out / shrink ran out of attempts
    const bool do_sum = p.sum != nullptr;
    long long held = do_sum ? p.held[cx.chain_ok ? cx.chain : 0] : 0;        // running sums in sojourn form (KParams::held)

    // An accepted proposal IS the next transition's current state: after an accept the lane's registers xp (value) and gp (gradient)
    // hold exactly what was just stored to X / GR, so the next transition starts from them and only a chain whose last proposal was
    // rejected (2 % of the transitions at cfg 3) re-reads its state — VERDICT r3 item 9 step 1: the kernel used to re-load x and g every
    // transition (2 x 800 B per chain and transition, 2.7 GB of the 6.3 GB a 32-transition launch of cfg 3 moved).
    double xp[NE], gp[NG];
    bool have = false;
    for (int s = 0; s < kl.nsteps; ++s) {
        const unsigned long long t = kl.t0 + (unsigned long long)s;
        if (cnt) tune_count_proposal(p, tn);
        bool acc = false;
        double ltp = lt;
        if (!have) mload<NE>(cx, p.X, p.D, xp);                        // current value

        if (SAMPLER == KLARA_SAMPLER_HMC) {
            // iterate/HMC.jl:124-201, leapfrog! samplers.jl:122-134
            double mom[NE], red[2];
            if (!have) {
                double g0[NE];
                mload<NE>(cx, p.GR, p.D, g0);                           // HMC.jl:140
#pragma unroll
                for (int e = 0; e < NE; ++e) gp[e] = g0[e];
            }
            mnormals<NE>(cx, p.seed, gch(), t, mom);                  // HMC.jl:135
            double k0[1] = { 0.0 };
#pragma unroll
            for (int e = 0; e < NE; ++e) k0[0] = k0[0] + mom[e] * mom[e];
            mreduce<1>(k0, cx.lane);
            const double H0 = lt - 0.5 * k0[0];                        // HMC.jl:137
            const double eps = tn.step, halfe = 0.5 * eps;
            // dual averaging: per-chain trip count (iterate/HMC.jl:142-144); the 16 chains of the tile run to the
            // longest trajectory, a finished chain's lanes re-use their frozen state (its MFMA columns are
            // recomputed but discarded)
            // leapfrog! L times (HMC.jl:146-155, samplers.jl:122-134) in its merged form — DESIGN.md section 2, deliberate deviation (7),
            // mirrored by the oracle for this layout: adjacent half-kicks are one update and every update is one fma, with the
            // gradient's sign folded in (the matrix cores return +P (x - mu)):
            //   p = fma(eps/2, g, p);  L x { x = fma(eps, p, x);  a = P (x - mu);  p = fma(-(l < L-1 ? eps : eps/2), a, p) };  g = -a
            // 2 NE instead of 6 NE + 4 MT vector instructions per gradient beside the MFMAs, which run on the same FP64 units.
#pragma unroll
            for (int e = 0; e < NE; ++e) mom[e] = kd_fma(halfe, gp[e], mom[e]);
            if (!da) {
                const int nl = p.nleaps;
                for (int l = 0; l < nl; ++l) {
#pragma unroll
                    for (int e = 0; e < NE; ++e) xp[e] = kd_fma(eps, mom[e], xp[e]);
                    dense_grad<NE, HASMU, false>(ldsP, cx.lane, xp, gp, ldsMu);
                    const double nkf = l + 1 < nl ? -eps : -halfe;
#pragma unroll
                    for (int e = 0; e < NE; ++e) mom[e] = kd_fma(nkf, gp[e], mom[e]);
                }
#pragma unroll
                for (int e = 0; e < NG; ++e) gp[e] = -gp[e];
            } else {
                const int nl = cx.chain_ok ? da_nleaps(p, eps) : 1;      // (a padding lane must not set the wavefront's trip count)
                const int nlmax = wave_max_int(nl);
                for (int l = 0; l < nlmax; ++l) {
                    const bool go = l < nl;
                    if (go) {
#pragma unroll
                        for (int e = 0; e < NE; ++e) xp[e] = kd_fma(eps, mom[e], xp[e]);
                    }
                    double gn[NG];
                    dense_grad<NE, HASMU, false>(ldsP, cx.lane, xp, gn, ldsMu);
                    if (go) {
                        const double nkf = l + 1 < nl ? -eps : -halfe;
#pragma unroll
                        for (int e = 0; e < NG; ++e) gp[e] = -gn[e];
#pragma unroll
                        for (int e = 0; e < NE; ++e) mom[e] = kd_fma(nkf, gn[e], mom[e]);
                    }
                }
            }
            double l1 = 0.0, k1 = 0.0;
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                l1 = l1 + dx(xp, e) * gp[e];                           // lt' = c + 1/2 (x'-mu).g'   (HMC.jl:157)
                k1 = k1 + mom[e] * mom[e];
            }
            red[0] = l1; red[1] = k1;
            mreduce<2>(red, cx.lane);
            ltp = p.gconst + 0.5 * red[0];
            const double H1 = ltp - 0.5 * red[1];                      // HMC.jl:159
            const double ratio = H1 - H0;                              // HMC.jl:161
            const double ex = kd_exp(ratio);
            const double a = 1.0 < ex ? 1.0 : ex;                      // HMC.jl:163
            const double u = kd_accept_uniform(kd_stream_block(p.seed, gch(), t, (uint32_t)((p.D + 1) >> 1)));
            acc = u < a;                                               // HMC.jl:165
            if (da) da_update(p, tn, (long long)t + 1, a);             // HMC.jl:225-249
        } else if (SAMPLER == KLARA_SAMPLER_MALA) {
            // iterate/MALA.jl:78-128
            double z[NE], xc[NE], red[3];
            const double h = tn.step, halfh = 0.5 * h, sq = __builtin_sqrt(h), inv_h = 1.0 / h, half_inv_h = 0.5 * inv_h;
            mnormals<NE>(cx, p.seed, gch(), t, z);
            double s1 = 0.0;
            {
                double g0[NE];
                if (have) {
#pragma unroll
                    for (int e = 0; e < NE; ++e) g0[e] = gp[e];
                } else {
                    mload<NE>(cx, p.GR, p.D, g0);
                }
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    xc[e] = xp[e];
                    const double mu = xc[e] + halfh * g0[e];           // MALA.jl:83
                    xp[e] = mu + sq * z[e];                            // MALA.jl:84
                    const double q1 = mu - xp[e];
                    s1 = s1 + (q1 * q1) * half_inv_h;                            // MALA.jl:90
                }
            }
            dense_grad<NE, HASMU>(ldsP, cx.lane, xp, gp, ldsMu);                     // MALA.jl:86
            double l1 = 0.0, s2 = 0.0;
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                l1 = l1 + dx(xp, e) * gp[e];
                const double mup = xp[e] + halfh * gp[e];              // MALA.jl:91
                const double q2 = mup - xc[e];
                s2 = s2 + (q2 * q2) * half_inv_h;                                // MALA.jl:92
            }
            red[0] = l1; red[1] = s1; red[2] = s2;
            mreduce<3>(red, cx.lane);
            ltp = p.gconst + 0.5 * red[0];
            double ratio = ltp - lt;                                   // MALA.jl:88
            ratio += red[1];
            ratio -= red[2];
            acc = ratio > 0.0;                                         // MALA.jl:94
            if (!acc && ratio > KD_LOG_UMIN_GUARD) {
                const double u = kd_accept_uniform(kd_stream_block(p.seed, gch(), t, (uint32_t)((p.D + 1) >> 1)));
                acc = ratio > kd_log_u01(u);
            }
        } else if (SAMPLER == KLARA_SAMPLER_SLICE) {
            // iterate/SliceSampler.jl:60-109.  Coordinate i = 4 e + q lives on lane q = i & 3 of its chain as register e = i >> 2.  A probe is a full
            // evaluation of the log-target: the gradient GEMM of the tile's 16 chains, then lt = c + 1/2 (x-mu).g.  Round 5: the chains of the tile
            // out of lockstep (slice_dense_free above) — every pass serves each chain's own next probe; it used to serve one stage of one
            // coordinate and wait for the slowest of the 16 at each of five call sites.
            double cur = lt;
            slice_dense_free<NE>(p, cx, gch(), t, xp, cur, stuck, [&](const double (&xt)[NE]) {
                double gt[NG], r1[1];
                dense_grad<NE, HASMU>(ldsP, cx.lane, xt, gt, ldsMu);
                double l1 = 0.0;
#pragma unroll
                for (int e = 0; e < NE; ++e) l1 = l1 + dx(xt, e) * gt[e];
                r1[0] = l1;
                mreduce<1>(r1, cx.lane);
                return p.gconst + 0.5 * r1[0];
            });
            ltp = cur;
            acc = true;                                                // the slice sampler always moves (SliceSampler.jl:108)
        } else {
            // iterate/MH.jl:72-124
            double z[NE], sg[NE], red[1];
            mnormals<NE>(cx, p.seed, gch(), t, z);
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                sg[e] = (4 * e + cx.q < p.D) ? p.vecparam[4 * e + cx.q] : 0.0;
                xp[e] = xp[e] + sg[e] * z[e];                          // MH.jl:79
            }
            dense_grad<NE, HASMU>(ldsP, cx.lane, xp, gp, ldsMu);                     // MH.jl:81
            double l1 = 0.0;
#pragma unroll
            for (int e = 0; e < NE; ++e) l1 = l1 + dx(xp, e) * gp[e];
            red[0] = l1;
            mreduce<1>(red, cx.lane);
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
            mstore<NE, KLARA_DENSE_COMMIT_AUX>(cx, p.X, p.D, xp);
            if (SAMPLER != KLARA_SAMPLER_MH && SAMPLER != KLARA_SAMPLER_SLICE) {
                double gs[NE];
#pragma unroll
                for (int e = 0; e < NE; ++e) gs[e] = gp[e];
                mstore<NE, KLARA_DENSE_COMMIT_AUX>(cx, p.GR, p.D, gs);
            }
            lt = ltp;
        }
        have = acc;                                      // (a rejected proposal leaves the registers holding the proposal: re-read next time)
        nacc += acc ? 1ull : 0ull;
        if (cnt && acc && SAMPLER != KLARA_SAMPLER_SLICE) tn.accepted += 1;       // (the slice sampler never counts accepts)
        if (accept_out != nullptr && cx.chain_ok && cx.q == 0)
            accept_out[(long long)s * p.nchains + cx.chain_here()] = acc ? 1 : 0;
        if (!PLAIN && !p.pooled && !da) tuning_block(p, tn);
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
            if (p.hist_lt != nullptr && col < p.hist_cols && cx.chain_ok && cx.q == 0)
                p.hist_lt[col * p.nchains + cx.chain_here()] = lt;
            if (SAMPLER != KLARA_SAMPLER_MH && SAMPLER != KLARA_SAMPLER_SLICE && p.hist_g != nullptr && col < p.hist_cols) {
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
    }

    if (SAMPLER == KLARA_SAMPLER_SLICE && stuck && cx.chain_ok && cx.q == 0) klara_raise(p.error_flag, KLARA_ERR_SLICE_STUCK);
    // (the lane index is re-read here — v_mbcnt — instead of keeping threadIdx.x & 63 alive through the launch for this one test: the register it
    // would occupy is the one the MFMA loop spills)
    const int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    if (lane_e < cx.here) {                                              // q == 0 (lanes 0..15) on an existing chain
        const long long chain_e = cx.first_chain + lane_e;
        p.LT[chain_e] = lt;
        p.naccept[chain_e] += nacc;
        if (do_sum) p.held[chain_e] = held;
        if (da) { p.da_epsbar[chain_e] = tn.epsbar; p.da_hbar[chain_e] = tn.hbar; }
        if (PLAIN) {
            // (nothing changed: tune_step[] still holds step0, the counters their zeros)
        } else if (!p.pooled) {
            p.tune_step[chain_e] = tn.step;
            p.tune_accepted[chain_e] = tn.accepted;
            p.tune_proposed[chain_e] = tn.proposed;
            p.tune_totproposed[chain_e] = tn.totproposed;
        } else if (cnt) {
            atomicAdd((unsigned long long*)p.pooled_accepted, (unsigned long long)tn.accepted - (unsigned long long)p.tune_accepted[0]);
        }
    }
}

// initialize! for the dense target: g = -P x, lt = c + 1/2 x.g, finiteness asserts
template <int NE, bool HASMU = false>
__global__ __launch_bounds__(512) void k_dense_init(const KParams p, const double* __restrict__ Pfrag, int needgrad)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MT = (NE + 3) / 4;
    double* ldsP = reinterpret_cast<double*>(smem);
    for (int i = threadIdx.x; i < MT * NE * 64 + (HASMU ? 4 * NE : 0); i += blockDim.x) ldsP[i] = Pfrag[i];
    const double* const ldsMu = ldsP + MT * NE * 64;
    __syncthreads();
    const MfmaCtx<NE> cx = make_mctx<NE>(p);
    double x[NE], g[4 * MT], red[1];
    mload<NE>(cx, p.X, p.D, x);
    dense_grad<NE, HASMU>(ldsP, cx.lane, x, g, ldsMu);
    double l1 = 0.0;
    bool bad = false;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        l1 = l1 + (HASMU ? x[e] - ldsMu[4 * e + cx.q] : x[e]) * g[e];
        if (needgrad) bad = bad || !kfinite(g[e]);
    }
    red[0] = l1;
    mreduce<1>(red, cx.lane);
    const double lt = p.gconst + 0.5 * red[0];
    bad = bad || (cx.chain_ok && !kfinite(lt));
    if (needgrad) {
        double gs[NE];
#pragma unroll
        for (int e = 0; e < NE; ++e) gs[e] = g[e];
        mstore<NE>(cx, p.GR, p.D, gs);
    }
    if (cx.chain_ok && cx.q == 0) p.LT[cx.chain] = lt;
    if (bad) klara_raise(p.error_flag, KLARA_ERR_NONFINITE_INIT);
}

#ifndef KLARA_DENSE_NO_PROBES     // (non-template kernels: one translation unit only)
// test hook: one v_mfma_f64_4x4x4_4b with per-lane operands (lane layout and accumulation order are pinned by tests)
__global__ void k_mfma_f64_4x4x4_probe(const double* A, const double* B, const double* C, double* Dout)
{
    const int lane = threadIdx.x & 63;
    Dout[lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[lane], B[lane], C[lane], 0, 0, 0);
}

// test hook: D[16x16] = A[16x4] * B[4x16] + C through one v_mfma_f64_16x16x4_f64, to pin the
// accumulation order of the instruction (one wave).
__global__ void k_mfma_f64_probe(const double* A, const double* B, const double* C, double* Dout)
{
    const int lane = threadIdx.x & 63;
    const double a = A[(lane & 15) * 4 + (lane >> 4)];        // A[i][k]
    const double b = B[(lane >> 4) * 16 + (lane & 15)];       // B[k][j]
    kd_double4 c;
#pragma unroll
    for (int r = 0; r < 4; ++r) c[r] = C[((lane >> 4) + 4 * r) * 16 + (lane & 15)];
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) Dout[((lane >> 4) + 4 * r) * 16 + (lane & 15)] = c[r];
}
#endif
