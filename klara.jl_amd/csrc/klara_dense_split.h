// klara_dense_split.h — the dense-Gaussian target beyond D = 256 on the FP64 matrix cores: a WORKGROUP carries the tile of 16 chains (layout kind 6).
//
// klara_dense_big.h gives one wavefront the whole vectors of its 16 chains: NE = 64 elements per lane at D = 256 is what 256 architectural + 256
// accumulator registers hold, at one wavefront per SIMD.  Beyond that neither the value nor the gradient fits a lane.  Here the W wavefronts of a
// workgroup SHARE the tile.  The gradient  G' = P X'  has MT = ceil(D / 16) row tiles of v_mfma_f64_16x16x4; W = 4 ceil(MT / (NEW / 4)) (4, 8, 12, 16: whole
// SIMD rounds; NEW = 16, 24 or 32 elements per lane and wavefront, below) and wavefront w owns the T = 2 .. NEW / 4 CONSECUTIVE tiles t0 .. t0 + T - 1 of an even deal (the first MT % W wavefronts one more than the
// rest: every SIMD carries the same number of tiles +- 1 whatever D is), i.e. the elements e = 4 t0 .. 4 (t0 + T) - 1 of every lane's column (lane
// (q, chain) of klara_dense.h: dimension i = 4 e + q) — the rows of its tiles — in the matrix pass and in every element-wise update (normals,
// proposal, kicks, sums).  What a wavefront needs from the others is the B operand of its pass, the whole proposal x of the 16 chains: every wavefront
// writes its elements to the LDS block xb[k-step = element][lane] (conflict-free 8-byte writes), a barrier, and the pass reads one 512-byte row per
// k-step (the mean is subtracted there).  The A fragments come from a k-major stream, ((kk * MT + t) * 64 + lane): the tiles of a wavefront are T x 512
// bytes in a row per k-step, taken through a ring of 2 T buffer loads with scalar offsets.  The sums of a transition (x.g, the proposal terms, the
// kinetic energy) are lane partials over the lane's elements in ascending order, the 4-lane tree (q0 + q1) + (q2 + q3) inside the wavefront, then the
// wavefronts' values in ascending order through LDS — the oracle's layout kind 6 (G = W).
//
// One kernel per sampler and NEW serves 257 <= D <= 1024 (every other count is a run-time value), at 2 .. 4 wavefronts per SIMD: the vector work of one (Box-Muller:
// 85 % of MALA's vector instructions) runs under the matrix passes of the others, which the 512-register kernels of klara_dense_big.h cannot do.
// The committed state lives in X / GR (written at every accept) and a transition reads what it needs from there, 8 elements at a time; the
// proposal is formed in the lane's own column of xb and found there again by the element-wise passes after the matrix pass.  Registers hold the
// pass's accumulators, its ring and, for HMC, the momentum.  MALA keeps the committed value in the LDS column and the committed gradient in the
// accumulators between transitions (no load between two transitions of a chain that accepts: RES below).
// MH, MALA, HMC (every tuner, dual averaging with per-chain trip counts), the slice sampler (the tile's 16 machines in every wavefront, the first one
// places candidates in xb), every monitor of the dense layouts.  Measurements and the steps that got here: profiles/r6_dense_split.txt.
#pragma once
#include "klara_dense.h"

// NEW (template parameter of everything below): elements per lane and wavefront, 4 per row tile a wavefront can own — 16, 24 or 32, whichever puts the fewest
// wavefronts on a tile (klara_launch.h klara_split_new: 4 wavefronts to D = 512, 8 beyond; the fewest wavefronts that hold the tile measured fastest)
#define KLARA_SPLIT_WMAX 16            // wavefronts per workgroup (D <= 1024)
#define KLARA_SPLIT_PAD 8              // k-steps of zeros behind the stream and rows behind the mean (>= R / 4 for every ring)
#ifndef KLARA_SPLIT_RESIDENT
#define KLARA_SPLIT_RESIDENT 1         // MALA / MH: the committed value stays in the lane's LDS column and MALA's committed gradient in the accumulators between transitions
#endif
#define KLARA_SPLIT_CH 8               // elements per group of loads (state, mean) in the element-wise passes

template <int NEW>
struct SplitCtx {
    MfmaCtx<NEW> m;        // the wavefront's <= 16 elements per lane as a 16-element column: offsets and validity shifted by its first element
    int w, W;                          // this wavefront, wavefronts per workgroup (scalars)
    int MT, t0, T;                     // row tiles of P; this wavefront's first tile and tile count (scalars); its elements: 4 t0 .. 4 (t0 + T) - 1
    int ksteps;                        // ceil(D / 4): the k-steps that are not all padding
    double* col;                       // LDS: this lane's column of xb, element e at col[e * 64]
    const double* xb;                  // LDS: the tile's proposal x, [k-step = element][lane]
    double* rbuf;                      // LDS: 2 x [value][wavefront][chain] partial sums (ping-pong)
    __amdgpu_buffer_rsrc_t wP;         // the fragment stream (+ the mean behind it)
    unsigned muoff;                    // byte offset of mu[0] in the stream
    unsigned par;                      // reduction parity
    // element e of the lane's 16-element column belongs to this wavefront (e < 4 T: a scalar test); the rows of xb behind it are the next wavefront's
    __device__ __forceinline__ bool own(int e) const { return (e >> 2) < T; }
    __device__ __forceinline__ double rd(int e) const { return own(e) ? col[e * 64] : 0.0; }
    __device__ __forceinline__ void wr(int e, double v) const { if (own(e)) col[e * 64] = v; }
};

template <int NEW>
__device__ __forceinline__ SplitCtx<NEW> make_sctx(const KParams& p, const double* Pfrag, char* smem)
{
    SplitCtx<NEW> s;
    MfmaCtx<NEW>& c = s.m;
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
    s.MT = (p.D + 15) >> 4;
    {
        const int base = s.MT / s.W, rem = s.MT - base * s.W;
        s.T = base + (s.w < rem ? 1 : 0);
        s.t0 = s.w * base + (s.w < rem ? s.w : rem);
    }
    const int eb = 4 * s.t0;
    int nv = c.chain_ok ? (p.D - c.q + 3) / 4 - eb : 0;
    c.nv = nv < 0 ? 0 : (nv > 4 * s.T ? 4 * s.T : nv);
    c.voff0 = (unsigned)((c.cl * p.D + c.q) * 8 + 32 * eb);
    const int NE = 4 * s.MT;                                      // rows of xb: every wavefront's elements
    double* const xb = reinterpret_cast<double*>(smem);
    s.xb = xb;
    s.col = xb + (size_t)eb * 64 + c.lane;
    s.rbuf = xb + (size_t)NE * 64;
    const unsigned long long a = (unsigned long long)Pfrag;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    s.ksteps = (p.D + 3) >> 2;
    s.muoff = (unsigned)(s.MT * (s.ksteps + KLARA_SPLIT_PAD)) * 512u;
    s.wP = __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0,
                                             __builtin_amdgcn_readfirstlane((int)s.muoff + 32 * (NE + KLARA_SPLIT_PAD)), 0x00020000);
    s.par = 0u;
    return s;
}
// (klara_split_waves / klara_split_lds_bytes: klara_launch.h — the host's launch planning and the launcher share them)
// mu of the lane's element e (HASMU; zero past D)
template <int NEW>
__device__ __forceinline__ double split_mu(const SplitCtx<NEW>& s, int e)
{
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(s.wP, (unsigned)s.m.q * 8u, s.muoff + (unsigned)(4 * s.t0 + e) * 32u, 0));
}

// acc[j] = tile t0 + j of +P (x - mu), from zero (j < T; the others stay zero): the K loop over the ceil(D / 4) rows of xb that are not padding, two
// k-steps per trip.  The A fragments go through a ring of 2 T buffer loads: a fragment is requested one trip (2 T x 64 matrix cycles) before its use and
// the wavefronts of a tile read DIFFERENT tiles of a k-step, so a request is served by the L2 — or, at D = 1024, where P is 8 MB, from beyond it —, not
// by a line a neighbour just brought into the L1: the 3 .. 4 wavefronts per SIMD cover that, a deeper ring measured slower (registers).  The B operand
// comes one k-step ahead from LDS.  The stream ends with KLARA_SPLIT_PAD k-steps of zeros and the mean with as many rows (a scalar offset is not
// range-checked): no guard on the last prefetch.
template <bool HASMU, int T, int NEW>
__device__ __forceinline__ void split_pass_t(const SplitCtx<NEW>& s, kd_double4 (&acc)[NEW / 4])
{
    const unsigned strideK = (unsigned)s.MT * 512u;               // bytes between k-steps: MT fragments
    const unsigned voff = (unsigned)s.m.lane * 8u, vq = (unsigned)s.m.q * 8u;
    unsigned soff = (unsigned)s.t0 * 512u;                        // this wavefront's tiles inside a k-step
    unsigned moff = s.muoff;
    double ring[2 * T];
#pragma unroll
    for (int i = 0; i < 2 * T; ++i)
        ring[i] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(s.wP, voff, soff + (unsigned)(i / T) * strideK + (unsigned)(i % T) * 512u, 0));
    const double* xl = s.xb + s.m.lane;
    const double* const xend = xl + (size_t)(4 * s.MT - 1) * 64;
    double bn = xl[0];
    if (HASMU) bn = bn - __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(s.wP, vq, moff, 0));
    for (int k0 = 0; k0 < s.ksteps; k0 += 2) {
        soff += 2u * strideK;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const double b = bn;
            xl = xl < xend ? xl + 64 : xl;                        // (the last k-step reads its own row again)
            moff += 32u;
            bn = xl[0];
            if (HASMU) bn = bn - __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(s.wP, vq, moff, 0));
#pragma unroll
            for (int j = 0; j < T; ++j) {
                const int i = T * kk + j;
                // (the matrix instruction first, then the load INTO the register it has just read: written the other way round the new value needs a
                // register of its own and the loop's back edge a copy per fragment behind a wait for every load of the trip)
                acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(ring[i], b, acc[j], 0, 0, 0);
                ring[i] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(s.wP, voff, soff + (unsigned)kk * strideK + (unsigned)j * 512u, 0));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}
template <bool HASMU, int NEW>
__device__ __forceinline__ void split_pass(const SplitCtx<NEW>& s, kd_double4 (&acc)[NEW / 4])
{
#pragma unroll
    for (int j = 0; j < NEW / 4; ++j) acc[j] = (kd_double4){ 0.0, 0.0, 0.0, 0.0 };
    switch (s.T) {                                                // (a scalar)
    case 8: if constexpr (NEW >= 32) split_pass_t<HASMU, 8>(s, acc); break;
    case 7: if constexpr (NEW >= 32) split_pass_t<HASMU, 7>(s, acc); break;
    case 6: if constexpr (NEW >= 24) split_pass_t<HASMU, 6>(s, acc); break;
    case 5: if constexpr (NEW >= 24) split_pass_t<HASMU, 5>(s, acc); break;
    case 4: split_pass_t<HASMU, 4>(s, acc); break;
    case 3: split_pass_t<HASMU, 3>(s, acc); break;
    case 2: split_pass_t<HASMU, 2>(s, acc); break;
    case 1: split_pass_t<HASMU, 1>(s, acc); break;
    default: break;                                               // (fewer tiles than wavefronts: nothing of the product is this wavefront's)
    }
}

// all-reduce of N sums over the tile's W wavefronts: 4-lane tree inside the wavefront, then the wavefronts in ascending order (oracle: layout kind 6).
// Its barrier also closes the matrix pass before it: once a wavefront is through, every wavefront of the tile has finished reading xb.
template <int N, int NEW>
__device__ __forceinline__ void split_reduce(SplitCtx<NEW>& s, double (&v)[N])
{
    static_assert(N <= 3, "rbuf holds three values");
    mreduce<N>(v, s.m.lane);
    double* const rb = s.rbuf + (size_t)(s.par & 1u) * 3 * s.W * 16;
    s.par ^= 1u;
    if (s.m.q == 0) {
#pragma unroll
        for (int i = 0; i < N; ++i) rb[(i * s.W + s.w) * 16 + s.m.cl] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double t = rb[(i * s.W) * 16 + s.m.cl];
        for (int w2 = 1; w2 < s.W; ++w2) t = t + rb[(i * s.W + w2) * 16 + s.m.cl];
        v[i] = t;
    }
}

// normals of the lane's elements, handed to f(e, z) in ascending e — consumed where they are drawn: no array of them.  mnormals of klara_dense.h for
// a column that starts at element 4 t0: the four elements of row tile G = t0 + g take their two Box-Muller pairs from half G & 1 of the blocks
// 8 (G >> 1) + {0, 4} + lane slot (words (x, y) for an even tile, (z, w) for an odd one: one pair of blocks per two tiles, formed at the even tile or at
// the wavefront's first).
template <int NEW, class F>
__device__ __forceinline__ void split_normals_each(const SplitCtx<NEW>& s, unsigned long long seed, unsigned long long gchain, unsigned long long t, F f)
{
    const MfmaCtx<NEW>& c = s.m;
    const uint32_t sh = (uint32_t)(c.q >> 1);
    const bool odd = (c.q & 1) != 0;
    const int nv = c.nv_here();
    const uint32_t lane_slot = (odd ? 2u : 0u) + sh;
    uint32_t st[4] = { 0u, 0u, 0u, 0u };
#pragma unroll
    for (int g = 0; g < NEW / 4; ++g) {
        if (g < s.T) {
            const int G = s.t0 + g;
            const bool half = (G & 1) != 0;
            uint32_t wa[2], wb[2];
            if (!half || g == 0) {
                const uint32_t slot = 8u * (uint32_t)(G >> 1) + lane_slot;
                const kd_u32x4 b0 = kd_stream_block(seed, gchain, t, slot), b1 = kd_stream_block(seed, gchain, t, slot + 4u);
                wa[0] = half ? b0.z : b0.x; wb[0] = half ? b0.w : b0.y; wa[1] = half ? b1.z : b1.x; wb[1] = half ? b1.w : b1.y;
                st[0] = b0.z; st[1] = b0.w; st[2] = b1.z; st[3] = b1.w;
            } else {
                wa[0] = st[0]; wb[0] = st[1]; wa[1] = st[2]; wb[1] = st[3];
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int e = 4 * g + 2 * h;
                double z0, z1, u1, lg;
                kd_normal_pair_w(wa[h], wb[h], &z0, &z1, &u1, &lg);
                const double recv = bperm_xor(odd ? z0 : z1, c.lane, 16);
                f(e, e < nv ? (odd ? recv : z0) : 0.0);               // even q: cos half of pair(e);   odd q: sin half of pair(e) from the partner
                f(e + 1, e + 1 < nv ? (odd ? z1 : recv) : 0.0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

// 8 elements e0 .. e0 + 7 of the lane's column of a (chains x D) array
template <int NEW>
__device__ __forceinline__ void split_load8(const MfmaCtx<NEW>& c, __amdgpu_buffer_rsrc_t w, int e0, int nv, double (&v)[KLARA_SPLIT_CH])
{
#pragma unroll
    for (int j = 0; j < KLARA_SPLIT_CH; ++j) v[j] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(w, c.off_fresh(e0 + j, nv), 0, 0));
}

// MW: wavefronts per SIMD the registers allow (3: 168 registers, workgroups of up to 6 wavefronts; 4: 128)
template <int SAMPLER, bool DA, bool HASMU, int MW, int NEW>
__global__ __launch_bounds__(256 * MW)
void k_dense_split(const KParams* __restrict__ pp, const KLaunch kl, const double* __restrict__ Pfrag)
{
    static_assert(SAMPLER == KLARA_SAMPLER_HMC || SAMPLER == KLARA_SAMPLER_MALA || SAMPLER == KLARA_SAMPLER_MH || SAMPLER == KLARA_SAMPLER_SLICE, "HMC, MALA, MH, slice");
    constexpr bool SLICE = SAMPLER == KLARA_SAMPLER_SLICE;
    static_assert(!DA || SAMPLER == KLARA_SAMPLER_HMC, "dual averaging is wired into HMC only (HMC.jl:124-133)");
    constexpr int NE = NEW, CH = KLARA_SPLIT_CH;
    constexpr bool NEEDG = SAMPLER != KLARA_SAMPLER_MH && !SLICE;
    const KParams& p = *pp;
    guchar* const accept_out = p.accept != nullptr ? p.accept + kl.t0 * (unsigned long long)p.nchains : nullptr;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    SplitCtx<NEW> sc = make_sctx<NEW>(p, Pfrag, smem);
    const MfmaCtx<NE>& cx = sc.m;
    kd_tables_to_lds();
    const bool w0 = sc.w == 0;
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
    bool stuck = false;                                  // slice sampler: step-out / shrink ran out of attempts
    const bool do_sum = p.sum != nullptr;
    long long held = do_sum ? p.held[cx.chain_ok ? cx.chain : 0] : 0;        // running sums in sojourn form (KParams::held)

    // The committed state lives in X / GR (written at every accept); a transition reads what it needs from there in groups of 8 elements, forms its
    // proposal in the lane's column of xb — where the matrix pass takes its B operand and the element-wise passes after it find the proposal again —
    // and keeps only the pass's accumulators (the proposal's gradient) and, for HMC, the momentum in registers.
    // MALA / MH (RES): no load from memory stands between two transitions of a chain that accepts.  An accepted proposal IS the next current value, and it
    // already sits in the lane's LDS column; its gradient already sits in the accumulators.  The current value a transition needs after its pass (MALA's
    // backward term; putting a rejected column back) is requested from X before the pass and arrives under it; only a lane that rejects re-reads its
    // gradient from GR, and that request goes out at the accept test, ahead of the commit and of the next transition's first Philox blocks.
    // (MH: measured slower in this form — 40.1 against 45.6 TFLOP/s at D = 320, 59.8 against 62.7 at 1,024: the scales and the requested value cost 64 registers, 270-400 B of scratch —,
    // so MH keeps reading value and scales where it draws; KLARA_SPLIT_RESIDENT=2 builds it.  With the value resident and the scales still read where they are
    // used: 42.1 / 53.7 / 58.8 against 45.6 / 55.8 / 62.4 at D = 320 / 512 / 1,024 — at MH's 0.79 acceptance nearly every wavefront puts a rejected column back.)
    constexpr bool RES = KLARA_SPLIT_RESIDENT != 0 && (SAMPLER == KLARA_SAMPLER_MALA || (SAMPLER == KLARA_SAMPLER_MH && KLARA_SPLIT_RESIDENT == 2));
    kd_double4 ga[NEW / 4];                                            // +P (x' - mu) of the lane's 16 elements: element e = ga[e >> 2][e & 3]
    double sg[SAMPLER == KLARA_SAMPLER_MH && RES ? NE : 1];           // MH: the proposal scales of the lane's elements (0 past D)
    if (RES) {
        const __amdgpu_buffer_rsrc_t wX = mwin<NE>(cx, p.X, 0, p.D), wG = mwin<NE>(cx, p.GR, 0, p.D);
        const int nv = cx.nv_here();
#pragma unroll
        for (int e0 = 0; e0 < NE; e0 += CH) {
            double xv[CH], gv[CH];
            split_load8(cx, wX, e0, nv, xv);
            if (NEEDG) split_load8(cx, wG, e0, nv, gv);
#pragma unroll
            for (int j = 0; j < CH; ++j) { sc.wr(e0 + j, xv[j]); if (NEEDG) ga[(e0 + j) >> 2][(e0 + j) & 3] = -gv[j]; }
        }
        if (SAMPLER == KLARA_SAMPLER_MH) {
            const __amdgpu_buffer_rsrc_t wS = __builtin_amdgcn_make_buffer_rsrc((void*)p.vecparam, 0, p.D * 8, 0x00020000);
            const unsigned so = (unsigned)(16 * sc.t0 + cx.q) * 8u;
#pragma unroll
            for (int e = 0; e < (SAMPLER == KLARA_SAMPLER_MH && RES ? NE : 1); ++e)
                sg[e] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(wS, so + 32u * (unsigned)e, 0, 0));       // (0 past D)
        }
    }
    for (int s = 0; s < kl.nsteps; ++s) {
        const unsigned long long t = kl.t0 + (unsigned long long)s;
        if (cnt) tune_count_proposal(p, tn);
        bool acc = false;
        double ltp = lt;
        const __amdgpu_buffer_rsrc_t wX = mwin<NE>(cx, p.X, 0, p.D), wG = mwin<NE>(cx, p.GR, 0, p.D);
        // lt' = c + 1/2 (x' - mu).g' with g' = -ga: the lane's part
        const auto lt_part = [&]() {
            double l1 = 0.0;
#pragma unroll
            for (int e0 = 0; e0 < NE; e0 += CH) {
                double xv[CH], mv[CH];
#pragma unroll
                for (int j = 0; j < CH; ++j) { xv[j] = sc.rd(e0 + j); mv[j] = HASMU ? split_mu(sc, e0 + j) : 0.0; }
#pragma unroll
                for (int j = 0; j < CH; ++j) l1 = l1 + (HASMU ? xv[j] - mv[j] : xv[j]) * -(double)ga[(e0 + j) >> 2][(e0 + j) & 3];
            }
            return l1;
        };

        if constexpr (SAMPLER == KLARA_SAMPLER_HMC) {
            // iterate/HMC.jl:124-201, leapfrog! samplers.jl:122-134 (merged fma form: DESIGN.md section 2 (7))
            double mom[NE];
#pragma unroll
            for (int e = 0; e < NE; ++e) mom[e] = 0.0;                   // (elements past the wavefront's tiles: no normals are drawn for them)
            const double eps = tn.step, halfe = 0.5 * eps;
            double k0[1] = { 0.0 };
            split_normals_each(sc, p.seed, gch(), t, [&](int e, double z) { mom[e] = z; k0[0] = k0[0] + z * z; });        // HMC.jl:135
            split_reduce<1>(sc, k0);                                   // (its barrier: nobody still reads xb)
            const double H0 = lt - 0.5 * k0[0];                        // HMC.jl:137
            {
                const int nv = cx.nv_here();
#pragma unroll
                for (int e0 = 0; e0 < NE; e0 += CH) {
                    double xv[CH], gv[CH];
                    split_load8(cx, wX, e0, nv, xv);                   // HMC.jl:139
                    split_load8(cx, wG, e0, nv, gv);                   // HMC.jl:140
#pragma unroll
                    for (int j = 0; j < CH; ++j) {
                        mom[e0 + j] = kd_fma(halfe, gv[j], mom[e0 + j]);
                        sc.wr(e0 + j, xv[j]);
                    }
                }
            }
            // dual averaging: per-chain trip count (iterate/HMC.jl:142-144); the tile runs to its longest trajectory (the same count in every
            // wavefront: each holds all 16 chains).  A finished chain's value and momentum stop changing, so the gradient the later passes recompute
            // for it is the one it already has, bit for bit: only the two updates are masked — by their STEP: fma(0, p, x) = x and fma(0, g, p) = p
            // exactly for finite p, g (one select per update instead of one per element and both versions of every element alive: the element-wise
            // selects cost this kernel 100 scratch accesses per leapfrog, matrix pipes 0.64 busy against the plain kernel's 0.86).
            const int nl = da ? (cx.chain_ok ? da_nleaps(p, eps) : 1) : p.nleaps;
            const int nlmax = da ? wave_max_int(nl) : nl;              // the tile's longest trajectory: a scalar trip count (a vote per trip cost the loop its registers)
            for (int l = 0; l < nlmax; ++l) {
                const bool go = !da || l < nl;
                const double eps_l = go ? eps : 0.0;
                if (l > 0) __syncthreads();                            // the previous pass is over in every wavefront: xb may change
#pragma unroll
                for (int e0 = 0; e0 < NE; e0 += CH) {
                    double xv[CH];
#pragma unroll
                    for (int j = 0; j < CH; ++j) xv[j] = sc.rd(e0 + j);
#pragma unroll
                    for (int j = 0; j < CH; ++j) sc.wr(e0 + j, kd_fma(eps_l, mom[e0 + j], xv[j]));
                }
                __syncthreads();
                split_pass<HASMU>(sc, ga);
                const double nkf = go ? (l + 1 < nl ? -eps : -halfe) : 0.0;
#pragma unroll
                for (int e = 0; e < NE; ++e) mom[e] = kd_fma(nkf, (double)ga[e >> 2][e & 3], mom[e]);
            }
            double red[2], k1 = 0.0;
#pragma unroll
            for (int e = 0; e < NE; ++e) k1 = k1 + mom[e] * mom[e];
            red[0] = lt_part(); red[1] = k1;                           // lt' = c + 1/2 (x'-mu).g'   (HMC.jl:157)
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
            const double h = tn.step, halfh = 0.5 * h, sq = __builtin_sqrt(h), half_inv_h = 0.5 * (1.0 / h);
            double s1 = 0.0;
            double xc[RES ? NE : 1];                                   // RES: the current value, from X, requested before the pass for use after it
            if constexpr (RES) {
                split_normals_each(sc, p.seed, gch(), t, [&](int e, double z) {
                    const double mu = sc.col[e * 64] + halfh * -(double)ga[e >> 2][e & 3];          // MALA.jl:83
                    const double xn = mu + sq * z;                     // MALA.jl:84
                    sc.col[e * 64] = xn;
                    const double q1 = mu - xn;
                    s1 = s1 + (q1 * q1) * half_inv_h;                  // MALA.jl:90
                });
                const int nv = cx.nv_here();
#pragma unroll
                for (int e = 0; e < (RES ? NE : 1); ++e) xc[e] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(wX, cx.off_fresh(e, nv), 0, 0));
            } else {
                const int nv = cx.nv_here();
                double xv[CH], gv[CH];
                split_load8(cx, wX, 0, nv, xv); split_load8(cx, wG, 0, nv, gv);
                split_normals_each(sc, p.seed, gch(), t, [&](int e, double z) {
                    const double mu = xv[e & (CH - 1)] + halfh * gv[e & (CH - 1)];          // MALA.jl:83
                    const double xn = mu + sq * z;                     // MALA.jl:84
                    sc.col[e * 64] = xn;
                    const double q1 = mu - xn;
                    s1 = s1 + (q1 * q1) * half_inv_h;                  // MALA.jl:90
                    if ((e & (CH - 1)) == CH - 1 && e + 1 < NE) { split_load8(cx, wX, e + 1, nv, xv); split_load8(cx, wG, e + 1, nv, gv); }
                });
            }
            __syncthreads();
            split_pass<HASMU>(sc, ga);                                 // MALA.jl:86
            double l1 = 0.0, s2 = 0.0;
            {
                const int nv = cx.nv_here();
#pragma unroll
                for (int e0 = 0; e0 < NE; e0 += CH) {
                    double xo[CH], xn[CH], mv[CH];
                    if (!RES) split_load8(cx, wX, e0, nv, xo);         // the current value (X holds the committed state)
#pragma unroll
                    for (int j = 0; j < CH; ++j) { xn[j] = sc.rd(e0 + j); mv[j] = HASMU ? split_mu(sc, e0 + j) : 0.0; }
#pragma unroll
                    for (int j = 0; j < CH; ++j) {
                        const double g = -(double)ga[(e0 + j) >> 2][(e0 + j) & 3];
                        l1 = l1 + (HASMU ? xn[j] - mv[j] : xn[j]) * g;
                        const double mup = xn[j] + halfh * g;          // MALA.jl:91
                        const double q2 = mup - (RES ? xc[RES ? e0 + j : 0] : xo[j]);
                        s2 = s2 + (q2 * q2) * half_inv_h;              // MALA.jl:92
                    }
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
            if constexpr (RES) {
                // a lane that rejects: the current value back into its column; its gradient from GR, negated, into the accumulators.  (First the
                // accepting lanes' gradient goes out — the commit below stores -ga, which a rejecting lane is about to overwrite: their stores are masked.)
                if (__any(!acc)) {
                    const int nv = cx.nv_here();
#pragma unroll
                    for (int e = 0; e < NE; ++e) {
                        const double gv = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(wG, cx.off_fresh(e, nv), 0, 0));
                        if (!acc) { sc.wr(e, xc[RES ? e : 0]); ga[e >> 2][e & 3] = -gv; }
                    }
                }
            }
        } else if constexpr (SLICE) {
            // iterate/SliceSampler.jl:60-109, the chains of the tile out of lockstep (slice_free_machine, klara_kernels.h): a probe is one matrix pass over the
            // tile's 16 chains, each at its own coordinate and stage.  Every wavefront runs the 16 little machines (the same values in all of them: the
            // probe's log-target comes out of the reduction); the first wavefront writes a chain's candidate into xb, where the pass and the owner of the
            // element find it.  A machine reads the value of its NEXT coordinate when it starts the current one: that slot is written by nobody until the
            // machine gets there, whereas the current one is being overwritten by the first wavefront while a slower one may not have read it yet.
            {
                const int nv = cx.nv_here();
#pragma unroll
                for (int e0 = 0; e0 < NE; e0 += CH) {
                    double xv[CH];
                    split_load8(cx, wX, e0, nv, xv);
#pragma unroll
                    for (int j = 0; j < CH; ++j) sc.wr(e0 + j, xv[j]);
                }
            }
            __syncthreads();
            double* const xbw = const_cast<double*>(sc.xb) + cx.cl;
            const auto slot = [&](int i) { return (i >> 2) * 64 + 16 * (i & 3); };
            int have_i = 0;
            double have_x = xbw[slot(0)], next_x = xbw[slot(p.D > 1 ? 1 : 0)];
            __syncthreads();                                           // (nobody places a candidate before everybody has these)
            double cur = lt;
            slice_free_machine(p, cx.chain_ok, gch(), t, cur, stuck,
                [&](int i, double& xs, double& ws) {
                    const bool adv = i != have_i;                      // the machine has moved on to coordinate have_i + 1
                    const int in = i + 1 < p.D ? i + 1 : i;
                    const double nx = xbw[slot(in)];
                    have_x = adv ? next_x : have_x;
                    next_x = adv ? nx : next_x;
                    have_i = i;
                    xs = have_x;
                    ws = p.vecparam[i];
                },
                [&](int i, bool on, double cand) { if (w0 && cx.q == 0 && on) xbw[slot(i)] = cand; },
                [&]() {
                    __syncthreads();
                    split_pass<HASMU>(sc, ga);
                    double r1[1] = { lt_part() };
                    split_reduce<1>(sc, r1);                           // (its barrier closes the pass: the next candidate may be placed)
                    return p.gconst + 0.5 * r1[0];
                });
            __syncthreads();                                           // (a stuck chain's coordinate was put back)
            ltp = cur;
            acc = true;                                                // the slice sampler always moves (SliceSampler.jl:108)
        } else {
            // iterate/MH.jl:72-124
            double red[1];
            double xc[RES ? NE : 1];
            if constexpr (RES) {
                split_normals_each(sc, p.seed, gch(), t, [&](int e, double z) { sc.col[e * 64] = sc.col[e * 64] + sg[RES ? e : 0] * z; });      // MH.jl:79
                const int nv = cx.nv_here();
#pragma unroll
                for (int e = 0; e < (RES ? NE : 1); ++e) xc[e] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(wX, cx.off_fresh(e, nv), 0, 0));
            } else {
                const int nv = cx.nv_here();
                const __amdgpu_buffer_rsrc_t wS = __builtin_amdgcn_make_buffer_rsrc((void*)p.vecparam, 0, p.D * 8, 0x00020000);
                const unsigned so = (unsigned)(16 * sc.t0 + cx.q) * 8u;
                double xv[CH], sgv[CH];
                const auto ld = [&](int e0) {
                    split_load8(cx, wX, e0, nv, xv);
#pragma unroll
                    for (int j = 0; j < CH; ++j) sgv[j] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(wS, so + 32u * (unsigned)(e0 + j), 0, 0));   // (0 past D)
                };
                ld(0);
                split_normals_each(sc, p.seed, gch(), t, [&](int e, double z) {
                    sc.col[e * 64] = xv[e & (CH - 1)] + sgv[e & (CH - 1)] * z;          // MH.jl:79
                    if ((e & (CH - 1)) == CH - 1 && e + 1 < NE) ld(e + 1);
                });
            }
            __syncthreads();
            split_pass<HASMU>(sc, ga);                                 // MH.jl:81
            red[0] = lt_part();
            split_reduce<1>(sc, red);
            ltp = p.gconst + 0.5 * red[0];
            const double ratio = ltp - lt;                             // MH.jl:83
            acc = ratio > 0.0;                                         // MH.jl:97
            if (!acc && ratio > KD_LOG_UMIN_GUARD) {
                const double u = kd_accept_uniform(kd_stream_block(p.seed, gch(), t, (uint32_t)((p.D + 1) >> 1)));
                acc = ratio > kd_log_u01(u);
            }
            if constexpr (RES) {
                if (__any(!acc)) {                                     // a lane that rejects: the current value back into its column
#pragma unroll
                    for (int e = 0; e < NE; ++e) { if (!acc) sc.wr(e, xc[RES ? e : 0]); }
                }
            }
        }

        const long long i1 = (long long)t + 1;
        const bool in_post = i1 > p.burnin && i1 <= p.nsteps_total;
        const bool save_now = in_post && sphase == 0;
        if (in_post) sphase = (sphase + 1 == (int)p.thinning) ? 0 : sphase + 1;
        const long long col = scol;
        if (save_now) scol += 1;
        const bool hist_now = save_now && col < p.hist_cols;
        // commit, fold and history, 8 elements at a time: the state being left from X, the proposal from xb, its gradient from the accumulators
        if (__any(acc) || (save_now && (p.hist != nullptr || (NEEDG && p.hist_g != nullptr)))) {
            const bool fold = do_sum && acc && held > 0;               // leaving a state after `held` saved steps: fold it into the sums
            const double hf = (double)held;
            const __amdgpu_buffer_rsrc_t ws = mwin<NE>(cx, p.sum, 0, p.D), wq = mwin<NE>(cx, p.sumsq, 0, p.D);
            const __amdgpu_buffer_rsrc_t wh = mwin<NE>(cx, p.hist, col * p.nchains, p.D), whg = mwin<NE>(cx, p.hist_g, col * p.nchains, p.D);
            const int nv = cx.nv_here();
            const bool need_old = (do_sum && __any(fold)) || (hist_now && p.hist != nullptr && !__all(acc));
#pragma unroll
            for (int e0 = 0; e0 < NE; e0 += CH) {
                double xo[CH], xn[CH];
                if (need_old) split_load8(cx, wX, e0, nv, xo);
#pragma unroll
                for (int j = 0; j < CH; ++j) xn[j] = sc.rd(e0 + j);
                if (fold) {
#pragma unroll
                    for (int j = 0; j < CH; ++j) {
                        const unsigned o = cx.off_fresh(e0 + j, nv);
                        const double sv = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(ws, o, 0, 0));
                        const double qv = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(wq, o, 0, 0));
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(kd_uint2, sv + hf * xo[j]), ws, o, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(kd_uint2, qv + hf * (xo[j] * xo[j])), wq, o, 0, 0);
                    }
                }
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    const unsigned o = cx.off_fresh(e0 + j, nv);
                    const double g = -(double)ga[(e0 + j) >> 2][(e0 + j) & 3];
                    if (acc) {
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(kd_uint2, xn[j]), wX, o, 0, 0);
                        if (NEEDG) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(kd_uint2, g), wG, o, 0, 0);
                    }
                    if (hist_now && p.hist != nullptr) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(kd_uint2, acc ? xn[j] : xo[j]), wh, o, 0, 0);
                    if (NEEDG && hist_now && p.hist_g != nullptr && acc) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(kd_uint2, g), whg, o, 0, 0);
                }
                if (NEEDG && hist_now && p.hist_g != nullptr && !__all(acc)) {        // a rejecting chain saves its committed gradient
                    double go_[CH];
                    split_load8(cx, wG, e0, nv, go_);
                    if (!acc) {
#pragma unroll
                        for (int j = 0; j < CH; ++j) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(kd_uint2, go_[j]), whg, cx.off_fresh(e0 + j, nv), 0, 0);
                    }
                }
            }
            if (fold) held = 0;
        }
        if (acc) lt = ltp;
        nacc += acc ? 1ull : 0ull;
        if (cnt && acc && !SLICE) tn.accepted += 1;                                // (the slice sampler never counts accepts)
        if (accept_out != nullptr && w0 && cx.chain_ok && cx.q == 0)
            accept_out[(long long)s * p.nchains + cx.chain_here()] = acc ? 1 : 0;
        if (!p.pooled && !da) tuning_block(p, tn);
        else if (da && cnt && tn.phase == 0 && (long long)t + 1 <= p.da_nadapt) {
            tn.totproposed += tn.proposed; tn.accepted = 0; tn.proposed = 0;
        }
        if (save_now) {
            if (do_sum) held += 1;
            if (p.hist_lt != nullptr && hist_now && w0 && cx.chain_ok && cx.q == 0)
                p.hist_lt[col * p.nchains + cx.chain_here()] = lt;
        }
        // (a chain's X / GR rows are written and read back by the wavefront that owns the elements: no hazard across wavefronts; the next
        // transition's writes to xb come after this one's last reduction, i.e. after every wavefront's pass)
    }

    if (SLICE && stuck && w0 && cx.chain_ok && cx.q == 0) klara_raise(p.error_flag, KLARA_ERR_SLICE_STUCK);
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
template <bool HASMU, int NEW>
__global__ __launch_bounds__(1024) void k_dense_split_init(const KParams p, const double* __restrict__ Pfrag, int needgrad)
{
    constexpr int NE = NEW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    SplitCtx<NEW> sc = make_sctx<NEW>(p, Pfrag, smem);
    const MfmaCtx<NE>& cx = sc.m;
    double x[NE], red[1];
    mload<NE>(cx, p.X, p.D, x);
#pragma unroll
    for (int e = 0; e < NE; ++e) sc.wr(e, x[e]);
    __syncthreads();
    kd_double4 ga[NEW / 4];
    split_pass<HASMU>(sc, ga);
    double l1 = 0.0, g[NE];
    bool bad = false;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        g[e] = -(double)ga[e >> 2][e & 3];
        l1 = l1 + (HASMU ? x[e] - split_mu(sc, e) : x[e]) * g[e];
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
