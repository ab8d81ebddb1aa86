// klara_diagt.h — "pair-transposed" transition kernels for the diagonal Gaussian target (layout kind 3).
//
// The reference evaluates one chain's iterate! as whole-vector Julia expressions (MH.jl:72-124, MALA.jl:78-128,
// HMC.jl:124-201).  For lt = c - sum_i w_i (x_i - mu_i)^2 every one of those expressions is elementwise except the
// three sums that enter the Metropolis ratio, so a chain needs only a handful of lanes: here Q lanes (Q = 8) share a
// chain and a wavefront carries 64/Q chains.  Lane q of a chain owns the element PAIRS P = p*Q + q, p = 0..NP-1
// (elements 2P, 2P+1): a pair is one Philox block / one Box-Muller evaluation (element i <- slot i>>1, cos for even i,
// sin for odd i — the same stream as every other layout) and one 16-byte access, and for fixed p the Q lanes of a chain
// touch Q*16 contiguous bytes.  Against the group layout (32 lanes x 4 elements at D = 100) the per-wavefront fixed
// work — cross-lane reductions, the accept test, addressing, the loop — is shared by 8 chains instead of 2, the
// reduction tree is 3 DPP steps instead of 5, and 89% instead of 78% of the lane slots hold real elements.
//
// Scope: every tuner (Vanilla — the jobs the throughput figures are quoted on —, AcceptanceRate per chain or pooled,
// DualAveraging for HMC); MH, MALA, HMC and the slice sampler; any monitor; 17 <= D <= 16*NP (odd D: the pair holding the last
// element has no second element)
// (accept mask, running sums, value / logtarget / gradlogtarget history).  Everything else runs on
// the group layout (klara_kernels.h).
// Sums are taken per lane in ascending element order and then over the Q lanes by an xor butterfly — the oracle mirrors
// this order for layout kind 3 (oracle/klara_oracle.c ko_reduce).
#pragma once
#include "klara_kernels.h"

#ifndef KLARA_DIAGT_Q
#define KLARA_DIAGT_Q 8                       // lanes per chain
#endif
#define KLARA_DIAGT_CPW (64 / KLARA_DIAGT_Q)  // chains per wavefront
// Keeps the instruction scheduler from hoisting every pair's Philox/Box-Muller ahead of the elementwise work (which
// costs ~60 VGPRs and an occupancy step): pairs are processed KLARA_DT_FENCE_EVERY at a time.
#ifndef KLARA_DT_FENCE_EVERY
#define KLARA_DT_FENCE_EVERY 2
#endif
#ifndef KLARA_DT_SLICE_WF
#define KLARA_DT_SLICE_WF 4   // wavefronts per SIMD requested for the slice sampler's unmonitored kernels with up to 8 pairs per lane (running sums in registers: 2)
#endif
#ifndef KLARA_DT_W1
#define KLARA_DT_W1 4
#endif
#ifndef KLARA_Q4_MALA_WF
#define KLARA_Q4_MALA_WF 3   // wavefronts per SIMD requested for the 4-lane MALA kernels with more than 8 pairs per lane
#endif
#ifndef KLARA_DT_WF
#define KLARA_DT_WF 2     // wavefronts per SIMD requested for the fused / monitored / tuned instantiations
#endif
#ifdef KLARA_DT_PERSISTENT
#define KLARA_DT_GROUP_LOOP for (long long grp = kl.group0 + wave0; grp < kl.group_end && grp * CPW < p.nchains; grp += nwaves)
#else
#define KLARA_DT_GROUP_LOOP const long long grp = kl.group0 + wave0; if (grp < kl.group_end && grp * CPW < p.nchains)
#endif
#define KLARA_DT_PAIR_FENCE(pi) do { if (KLARA_DT_FENCE_EVERY > 0 && ((pi) + 1) % KLARA_DT_FENCE_EVERY == 0) __builtin_amdgcn_sched_barrier(0); } while (0)

// Per-launch "which kernel runs" protocol for jobs whose running sums can be kept two ways (untuned MH / MALA, 17 <= D <= 104):
// the 4-lane kernels fold the sums of a chain that moves straight into memory (atomic adds: nothing resident, cheapest while chains
// move rarely), the 8-lane kernels keep the sums of the chains that moved in registers (flat cost at any acceptance).  Both produce
// the same bits (the 4-lane kernels sum in the 8-lane order, see below), so the choice is a pure performance decision and is taken
// ON THE DEVICE, launch by launch: the host enqueues both kernels of a launch, each reads the decision cell of this launch
// (cell_in) and returns at once unless it names its own mode; the one that runs counts the accepted proposals of the launch and
// its last wavefront writes the decision for the NEXT launch (cell_out — the cells alternate, so the sibling of a launch still sees
// the value the launch was decided with).  No host round trip, no synchronisation, any number of launches in flight.
struct KAuto {
    const int* cell_in;             // decision this launch is subject to (nullptr: run unconditionally)
    int* cell_out;                  // decision for the next launch of this chain partition (nullptr: none kept)
    unsigned long long* acc_ctr;    // launch-wide (waves done << 40 | accepted proposals) counter, zero between launches
    int* mirror;                    // host-visible {mode, accepted of the launch (saturated), launch index, 0} per partition, or nullptr
    unsigned long long thr_work;    // accepted * 65536 > thr_work  ->  next launch keeps resident sums (mode 1)
    int my_mode;                    // 0: sums folded into memory (4 lanes per chain); 1: resident sums (8 lanes per chain)
    int fanout;                     // cells (stride 2 ints) / mirrors (stride 4) the decision is written to: 1, or 4 when the launch covers every partition
    int launch_idx;                 // index of this launch in the job (mirror[2]: tells the host how fresh the decision is)
};
#define KLARA_AUTO_NONE KAuto{ nullptr, nullptr, nullptr, nullptr, 0ull, 0, 1, 0 }

__device__ __forceinline__ unsigned wave_sum_u32(unsigned v)
{
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v += (unsigned)__builtin_amdgcn_ds_bpermute((int)((((unsigned)threadIdx.x & 63u) ^ (unsigned)m) << 2), (int)v);
    return v;
}
// end of a launch: every wavefront adds its accepted proposals to its workgroup's LDS counter, the last wavefront of a workgroup
// adds the workgroup's total to the launch counter (one returning atomic per workgroup: thousands of wavefronts finishing together
// on ONE address cost ~25 us per launch), and the last workgroup to arrive decides the next launch
__shared__ unsigned klara_auto_wg[2];      // {accepted, wavefronts done} of this workgroup; zeroed by auto_begin
__device__ __forceinline__ void auto_begin()           // before the kernel's first workgroup barrier
{
    if (threadIdx.x < 2) klara_auto_wg[threadIdx.x] = 0u;
}
__device__ __forceinline__ void auto_finish(const KAuto& ka, unsigned wave_acc)
{
    if (ka.acc_ctr == nullptr) return;
    const unsigned tot = wave_sum_u32(wave_acc);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&klara_auto_wg[0], tot);                                        // (LDS; in order with the next one)
        if (atomicAdd(&klara_auto_wg[1], 1u) != (blockDim.x >> 6) - 1) return;    // not the workgroup's last wavefront
        const unsigned long long wg_tot = atomicAdd(&klara_auto_wg[0], 0u);
        const unsigned long long old = atomicAdd(ka.acc_ctr, (1ull << 40) | wg_tot);
        if ((old >> 40) == (unsigned long long)gridDim.x - 1) {
            const unsigned long long total = (old & ((1ull << 40) - 1)) + wg_tot;
            *ka.acc_ctr = 0ull;
            const int mode = (total << 16) > ka.thr_work ? 1 : 0;
            if (ka.cell_out != nullptr) for (int k = 0; k < ka.fanout; ++k) ka.cell_out[2 * k] = mode;
            if (ka.mirror != nullptr) {
                for (int k = 0; k < ka.fanout; ++k) {
                    __hip_atomic_store(ka.mirror + 4 * k + 1, (int)(total > 0x7fffffffull ? 0x7fffffffull : total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(ka.mirror + 4 * k, mode, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(ka.mirror + 4 * k + 2, ka.launch_idx, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);   // (last: the host reads it first)
                }
            }
        }
    }
}

template <int NP, int Q>
struct PairCtx {
    int lane, q, cw;        // lane in the wavefront, lane within the chain, chain within the wavefront
    unsigned off0;          // byte offset of pair p = 0 inside the wavefront's chain window; pair p adds p*Q*16
    bool last_ok;           // the last pair (p = NP-1) holds real elements (all earlier pairs always do)
    bool last_full;         // ... both of them (odd D: the pair that holds element D-1 has no second element)
};

template <int NP, int Q>
__device__ __forceinline__ PairCtx<NP, Q> make_pctx(int D)
{
    PairCtx<NP, Q> c;
    c.lane = threadIdx.x & 63;
    c.q = c.lane & (Q - 1);
    c.cw = c.lane / Q;
    c.off0 = (unsigned)((c.cw * D + 2 * c.q) * 8);
    c.last_ok = 2 * ((NP - 1) * Q + c.q) < D;
    c.last_full = 2 * ((NP - 1) * Q + c.q) + 1 < D;
    return c;
}
template <int NP, int Q>
__device__ __forceinline__ unsigned pair_off(const PairCtx<NP, Q>& c, int p)
{
    const unsigned o = c.off0 + (unsigned)(p * Q * 16);
    return (p == NP - 1 && !c.last_ok) ? KLARA_BUF_OOB : o;
}
template <int NP, int Q>
__device__ __forceinline__ void load_pairs(const PairCtx<NP, Q>& c, __amdgpu_buffer_rsrc_t w, double (&v)[2 * NP])
{
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const kd_uint4 t = __builtin_amdgcn_raw_buffer_load_b128(w, pair_off<NP, Q>(c, p), 0, 0);
        v[2 * p] = __builtin_bit_cast(double, kd_uint2{ t.x, t.y });
        v[2 * p + 1] = __builtin_bit_cast(double, kd_uint2{ t.z, t.w });
        if (p == NP - 1 && !c.last_full) v[2 * p + 1] = 0.0;     // (odd D: those 8 bytes belong to the next chain)
    }
}
template <int NP, int Q>
__device__ __forceinline__ void store_pairs(const PairCtx<NP, Q>& c, __amdgpu_buffer_rsrc_t w, const double (&v)[2 * NP])
{
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const kd_uint2 a = __builtin_bit_cast(kd_uint2, v[2 * p]), b = __builtin_bit_cast(kd_uint2, v[2 * p + 1]);
        if (p < NP - 1) {
            __builtin_amdgcn_raw_buffer_store_b128(kd_uint4{ a.x, a.y, b.x, b.y }, w, pair_off<NP, Q>(c, p), 0, 0);
        } else {                                                 // the last pair may be half a pair (odd D): two 8-byte stores
            const unsigned o = pair_off<NP, Q>(c, p);
            __builtin_amdgcn_raw_buffer_store_b64(a, w, o, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b64(b, w, c.last_full ? o + 8u : KLARA_BUF_OOB, 0, 0);
        }
    }
}
// the same accesses for the lanes with `on` only (the others read zeros / store nothing: their offsets fall outside the window)
template <int NP, int Q>
__device__ __forceinline__ void load_pairs_if(const PairCtx<NP, Q>& c, __amdgpu_buffer_rsrc_t w, bool on, double (&v)[2 * NP])
{
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const kd_uint4 t = __builtin_amdgcn_raw_buffer_load_b128(w, on ? pair_off<NP, Q>(c, p) : KLARA_BUF_OOB, 0, 0);
        v[2 * p] = __builtin_bit_cast(double, kd_uint2{ t.x, t.y });
        v[2 * p + 1] = __builtin_bit_cast(double, kd_uint2{ t.z, t.w });
        if (p == NP - 1 && !c.last_full) v[2 * p + 1] = 0.0;
    }
}
template <int NP, int Q>
__device__ __forceinline__ void store_pairs_if(const PairCtx<NP, Q>& c, __amdgpu_buffer_rsrc_t w, bool on, const double (&v)[2 * NP])
{
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const kd_uint2 a = __builtin_bit_cast(kd_uint2, v[2 * p]), b = __builtin_bit_cast(kd_uint2, v[2 * p + 1]);
        const unsigned o = on ? pair_off<NP, Q>(c, p) : KLARA_BUF_OOB;
        if (p < NP - 1) {
            __builtin_amdgcn_raw_buffer_store_b128(kd_uint4{ a.x, a.y, b.x, b.y }, w, o, 0, 0);
        } else {
            __builtin_amdgcn_raw_buffer_store_b64(a, w, o, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b64(b, w, (on && c.last_full) ? o + 8u : KLARA_BUF_OOB, 0, 0);
        }
    }
}
// A chain leaves the state x after `held` saved steps (KParams::held): sum += held * x, sumsq += held * (x * x), held = 0, for
// the lanes with `fold`; their chain's sums are fetched first if this is the first time in the launch.  (Behind a wave-uniform
// branch at the call site: it runs on a few per cent of the transitions of the headline job.)
template <int NP, int Q>
__device__ __forceinline__ void diagt_fold(const PairCtx<NP, Q>& c, __amdgpu_buffer_rsrc_t wsum, __amdgpu_buffer_rsrc_t wsq, bool fold,
                                        bool& loaded, long long& held, const double (&x)[2 * NP], double (&sm)[2 * NP], double (&sq)[2 * NP])
{
    const bool need = fold && !loaded;
    if (__any(need)) {
        double ts[2 * NP], tq[2 * NP];
        load_pairs_if<NP, Q>(c, wsum, need, ts);
        load_pairs_if<NP, Q>(c, wsq, need, tq);
#pragma unroll
        for (int e = 0; e < 2 * NP; ++e) { sm[e] = need ? ts[e] : sm[e]; sq[e] = need ? tq[e] : sq[e]; }
        loaded = loaded || need;
    }
    const double hf = fold ? (double)held : 0.0;
#pragma unroll
    for (int e = 0; e < 2 * NP; ++e) {
        const double a = sm[e] + hf * x[e], b = sq[e] + hf * (x[e] * x[e]);
        sm[e] = fold ? a : sm[e]; sq[e] = fold ? b : sq[e];
    }
    held = fold ? 0 : held;
}

// The same fold without resident sums, as a plain read-modify-write of the chain's sums in memory (a chain's sums are only ever
// touched by its own lanes, in program order, and a launch never overlaps another launch of the same chains): 16-byte loads and
// stores of the moving chains' pairs, KLARA_DT_RMW_CHUNK pairs of sum and sumsq in flight at a time (8 transient registers per
// pair; the Philox / Box-Muller temporaries are dead here).  Used by the 4-lanes-per-chain kernels (untuned MH / MALA), which have
// no registers for resident sums.  (Round 2 folded with no-return FP64 atomic adds: nothing to wait for, but one 8-byte atomic
// per lane at the L2's atomic rate — 65 G/s — against 16 bytes per lane at the L2 / HBM rate: same-box, us per transition of
// 65,536 x 100 at 0.9 / 4 / 10 / 21 / 56 % acceptance: atomics 15.8 / 16.9 / 27 / 50 / 124, read-modify-write 15.3 / 16.6 / 17.5 / 19.1 / 22.3.)
#ifndef KLARA_DT_RMW_CHUNK
#define KLARA_DT_RMW_CHUNK 3
#endif
// The same fold as fire-and-forget atomic adds (global_atomic_add_f64 without return): sum += held x, sumsq += held x^2 element by element.
// Each element of a chain's sums is only ever touched by its own lane, one fold after the other, so the memory sees exactly the additions the
// read-modify-write form makes (one IEEE add of the rounded product per fold: the same bits) — but the wavefront does not wait for 2 NP loads
// before it can go on: at the 7 % acceptance of a fresh drift-0.9 job the load round trips of the folds were 1.5-2 of the 16.6 us per transition
// of all chains (h = 0.6, ~10 % acceptance: 15.9 -> see profiles/r4_ab_fold_atomic.txt).
template <int NP, int Q>
__device__ __forceinline__ void diagt_fold_atomic(const PairCtx<NP, Q>& c, gdouble* sum0, gdouble* sq0, bool fold, long long& held, const double (&x)[2 * NP])
{
    typedef __attribute__((address_space(1))) double* gptr;
    if (fold) {
        const double hf = (double)held;
        const gptr ps = (gptr)((__attribute__((address_space(1))) char*)sum0 + c.off0), pq = (gptr)((__attribute__((address_space(1))) char*)sq0 + c.off0);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const bool ok0 = p < NP - 1 || c.last_ok, ok1 = p < NP - 1 || c.last_full;
            if (ok0) {
                (void)__builtin_amdgcn_global_atomic_fadd_f64(ps + p * Q * 2, hf * x[2 * p]);
                (void)__builtin_amdgcn_global_atomic_fadd_f64(pq + p * Q * 2, hf * (x[2 * p] * x[2 * p]));
            }
            if (ok1) {
                (void)__builtin_amdgcn_global_atomic_fadd_f64(ps + p * Q * 2 + 1, hf * x[2 * p + 1]);
                (void)__builtin_amdgcn_global_atomic_fadd_f64(pq + p * Q * 2 + 1, hf * (x[2 * p + 1] * x[2 * p + 1]));
            }
        }
    }
    held = fold ? 0 : held;
}
template <int NP, int Q>
__device__ __forceinline__ void diagt_fold_rmw(const PairCtx<NP, Q>& c, __amdgpu_buffer_rsrc_t wsum, __amdgpu_buffer_rsrc_t wsq, bool fold,
                                               long long& held, const double (&x)[2 * NP])
{
    const double hf = fold ? (double)held : 0.0;
    constexpr int CH = KLARA_DT_RMW_CHUNK;
#pragma unroll
    for (int p0 = 0; p0 < NP; p0 += CH) {
        kd_uint4 ts[CH], tq[CH];
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            if (p0 + k < NP) {
                const unsigned o = fold ? pair_off<NP, Q>(c, p0 + k) : KLARA_BUF_OOB;
                ts[k] = __builtin_amdgcn_raw_buffer_load_b128(wsum, o, 0, 0);
                tq[k] = __builtin_amdgcn_raw_buffer_load_b128(wsq, o, 0, 0);
            }
        }
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            if (p0 + k < NP) {
                const int p = p0 + k;
                const unsigned o = fold ? pair_off<NP, Q>(c, p) : KLARA_BUF_OOB;
                const double s0 = __builtin_bit_cast(double, kd_uint2{ ts[k].x, ts[k].y }) + hf * x[2 * p];
                const double s1 = __builtin_bit_cast(double, kd_uint2{ ts[k].z, ts[k].w }) + hf * x[2 * p + 1];
                const double q0 = __builtin_bit_cast(double, kd_uint2{ tq[k].x, tq[k].y }) + hf * (x[2 * p] * x[2 * p]);
                const double q1 = __builtin_bit_cast(double, kd_uint2{ tq[k].z, tq[k].w }) + hf * (x[2 * p + 1] * x[2 * p + 1]);
                const kd_uint2 a = __builtin_bit_cast(kd_uint2, s0), b = __builtin_bit_cast(kd_uint2, s1);
                const kd_uint2 e = __builtin_bit_cast(kd_uint2, q0), f = __builtin_bit_cast(kd_uint2, q1);
                if (p < NP - 1) {
                    __builtin_amdgcn_raw_buffer_store_b128(kd_uint4{ a.x, a.y, b.x, b.y }, wsum, o, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(kd_uint4{ e.x, e.y, f.x, f.y }, wsq, o, 0, 0);
                } else {                                 // the last pair may be half a pair (odd D): 8-byte stores, the second only if it exists
                    const unsigned o1 = (fold && c.last_full) ? o + 8u : KLARA_BUF_OOB;
                    __builtin_amdgcn_raw_buffer_store_b64(a, wsum, o, 0, 0); __builtin_amdgcn_raw_buffer_store_b64(b, wsum, o1, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(e, wsq, o, 0, 0); __builtin_amdgcn_raw_buffer_store_b64(f, wsq, o1, 0, 0);
                }
            }
        }
    }
    held = fold ? 0 : held;
}

// per-element parameter vector (weights, means, proposal scales): element 2P+h of the lane's pair p
template <int NP, int Q>
__device__ __forceinline__ void load_pair_param(const PairCtx<NP, Q>& c, const gdouble* base, int D, double dflt, double (&v)[2 * NP])
{
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int i = 2 * (p * Q + c.q);
        v[2 * p] = dflt; v[2 * p + 1] = dflt;
        if (base != nullptr && i < D) v[2 * p] = base[i];
        if (base != nullptr && i + 1 < D) v[2 * p + 1] = base[i + 1];
    }
}

// proposal normals of pair slot p of the lane (pair index p*Q + q) for transition t; padding pairs get z = 0 (their x, g, parameters are
// 0 / defaults, so every term they contribute is exactly 0).  One Philox block serves pair indices i and i + 8 (detmath.h
// kd_normal_pair_at); with Q <= 8 lanes per chain both sit in the same lane, STEP = 8/Q slots apart: the block is formed at the first
// of the two and its words (z, w) wait in two registers (`stash`) for the second — 7 blocks per lane and transition instead of 13 in
// the 4-lane headline kernel.  With 16 / 32 lanes per chain the partner sits in another lane: every lane forms its block and takes
// its half.  (u1, log u1) are handed back: the last slot's are the accept draw when pair index ceil(D/2) falls on it (padding pairs
// take words (x, y) of block slot = pair index; see AccDraw in klara_kernels.h) — which needs the last slot to form a block of its own.
template <int Q> struct PairShare {
    static constexpr int STEP = Q <= 8 ? 8 / Q : 0;
    static constexpr bool second(int p) { return STEP != 0 && ((p / (STEP ? STEP : 1)) & 1) != 0; }
};
template <int NP, int Q>
__device__ __forceinline__ void pair_normals(const PairCtx<NP, Q>& c, unsigned long long seed, unsigned long long gchain,
                                             unsigned long long t, int p, uint32_t (&stash)[4], double& z0, double& z1, double& u1, double& lg1)
{
    constexpr int STEP = PairShare<Q>::STEP, S1 = STEP ? STEP : 1;
    uint32_t wa, wb;
    if (PairShare<Q>::second(p)) {
        wa = stash[2 * (p % S1)]; wb = stash[2 * (p % S1) + 1];
    } else {
        const uint32_t idx = (uint32_t)(p * Q + c.q);
        const bool pad = p == NP - 1 && !c.last_ok;
        const kd_u32x4 b = kd_stream_block(seed, gchain, t, pad ? idx : kd_pair_block(idx));
        if constexpr (STEP != 0) {                         // first of the two: half 0 by construction
            wa = b.x; wb = b.y;
            stash[2 * (p % S1)] = b.z; stash[2 * (p % S1) + 1] = b.w;
        } else {
            const bool hb = !pad && kd_pair_half(idx) != 0u;
            wa = hb ? b.z : b.x; wb = hb ? b.w : b.y;
        }
    }
    kd_normal_pair_w(wa, wb, &z0, &z1, &u1, &lg1);
    KLARA_PIN(z0); KLARA_PIN(z1);        // formed HERE: left to itself the compiler sinks the transforms of the stashed halves towards their uses (+60 registers)
    if (p == NP - 1 && !c.last_ok) z0 = 0.0;
    if (p == NP - 1 && !c.last_full) z1 = 0.0;
}

// compile-time loop: f(kd_int<I>{}) for I = FROM .. TO-1 (a register-array index that must stay a constant through nested run-time loops)
template <int V> struct kd_int { static constexpr int value = V; };
template <int FROM, int TO, class F>
__device__ __forceinline__ void kd_static_for(F&& f)
{
    if constexpr (FROM < TO) { f(kd_int<FROM>{}); kd_static_for<FROM + 1, TO>(f); }
}

// The diagonal target on one element (klara_kernels.h DiagTarget, same operations in the same order).  UNITW: w = 1 and
// mu = 0 (README.md:23 -dot(z,z)): x - 0, 1*(.) and (-2*1)*(.) are exact, so dropping them changes no bit.
// m2w = -2.0 * w, formed once per workgroup (the same product DiagTarget forms per evaluation).
template <bool UNITW>
__device__ __forceinline__ void diag_elem(double x, double w, double m2w, double m, double& term, double& grad)
{
    const double dd = UNITW ? x : x - m;
    term = UNITW ? dd * dd : w * (dd * dd);
    grad = UNITW ? -2.0 * dd : m2w * dd;
}

// ONESTEP: exactly one transition per launch and no saved-sample monitor (the accepted proposal goes straight from its
// registers to HBM).  MON: the save rule of BasicMCJob.jl:226-231 runs after every transition — per-chain running sums,
// value / logtarget / gradlogtarget history — on the committed state (never together with ONESTEP).
// TUNE: the tuner bookkeeping of the group-layout kernel (proposal / accept counters, AcceptanceRateMCTuner per chain or
// pooled per GPU, verbose counting) — the same device functions, per-chain state in registers over the launch.
// DA (HMC only): DualAveragingMCTuner — per-chain step and trajectory length (iterate/HMC.jl:142-144, 225-249); the
// wavefront runs to the longest trajectory of its chains, a finished chain's lanes keep their state.
template <int SAMPLER, int NP, int Q, bool ONESTEP, bool TUNE>
__host__ __device__ constexpr int diagt_min_waves()       // wavefronts per SIMD the register allocator is asked to leave room for
{
    return NP <= 8 ? (ONESTEP && SAMPLER != KLARA_SAMPLER_HMC ? KLARA_DT_W1 : KLARA_DT_WF)
                   : (Q == 4 && NP <= 13 && !TUNE ? (SAMPLER == KLARA_SAMPLER_MALA ? KLARA_Q4_MALA_WF : 2) : 1);
}

// USERPAIR (run-time compiled instantiations only, klara_custom_pair.h): the target is the user's pair closure
//     lt(x) = sum over element pairs P of klara_user_pair(x[2P], x[2P+1], P, ...)      (it also returns the pair's two partial derivatives)
// instead of the diagonal Gaussian — the device form of BasicContMuvParameter(:p, logtarget=f, gradlogtarget=g)
// (BasicContMuvParameter.jl:174-201) for targets that are sums of terms of one or two neighbouring coordinates, on the same few
// lanes per chain, the same stream, samplers, tuners, monitors and save rule.  The kernel's sums run over -term (lt = 0 - sum(-term):
// the same bits as the sum itself) and a pair contributes ONE term to them.
#ifdef KLARA_USER_PAIR_TARGET
#define KLARA_PAIR_CALL(a, b, P, g0, g1) klara_user_pair((a), (b), (P), D, (const double*)p.cdata, (long long)p.cndata, (g0), (g1))
#else
#define KLARA_PAIR_CALL(a, b, P, g0, g1) 0.0
#endif
template <int SAMPLER, int NP, int Q, bool ONESTEP, bool UNITW, bool MON, bool TUNE = false, bool DA = false, bool USERPAIR = false>
__global__ __launch_bounds__(256, (SAMPLER == KLARA_SAMPLER_SLICE && NP <= 8 && !MON && !TUNE ? KLARA_DT_SLICE_WF : diagt_min_waves<SAMPLER, NP, Q, ONESTEP, TUNE>()))
void k_diagt(const KParams* __restrict__ pp, const KLaunch kl, const KAuto ka)
{
#ifndef KLARA_USER_PAIR_TARGET
    static_assert(!USERPAIR, "pair closures exist in run-time compiled translation units only");
#endif
    static_assert(!USERPAIR || (UNITW && Q >= 8), "pair closures: 8 or more lanes per chain");
    if (ka.cell_in != nullptr && *ka.cell_in != ka.my_mode) return;       // (launch-uniform: the sibling kernel runs this launch)
    static_assert(!(ONESTEP && (MON || TUNE)), "monitored / tuned jobs run the committing kernel");
    static_assert(!DA || (TUNE && SAMPLER == KLARA_SAMPLER_HMC), "dual averaging: tuned HMC");
    constexpr bool PLAIN = !TUNE;              // KCNT / KPOOLED (klara_kernels.h) fold to 0 when nothing counts
    constexpr int E = 2 * NP, CPW = 64 / Q;
    constexpr bool NEEDG = SAMPLER == KLARA_SAMPLER_MALA || SAMPLER == KLARA_SAMPLER_HMC;
    constexpr bool SLICE = SAMPLER == KLARA_SAMPLER_SLICE;
    static_assert(!(SLICE && ONESTEP), "the slice sampler moves every chain: it runs the committing kernel");
    // running sums folded into memory instead of being held in registers (diagt_fold_rmw): untuned MH / MALA on the
    // 4-lanes-per-chain form of the layout
    constexpr bool MEMSUM = MON && !TUNE && Q == 4 && (SAMPLER == KLARA_SAMPLER_MH || SAMPLER == KLARA_SAMPLER_MALA);
    const KParams& p = *pp;
    // weights and means of a non-unit diagonal: one LDS copy per workgroup (element i at [i], padding = (1, 0)) instead
    // of 4*NP registers per lane; a pair's (w, mu) values are 16-byte LDS reads where they are used
    __shared__ __attribute__((aligned(16))) double lds_w[UNITW ? 2 : 2 * NP * Q];
    __shared__ __attribute__((aligned(16))) double lds_mu[UNITW ? 2 : 2 * NP * Q];
    __shared__ __attribute__((aligned(16))) double lds_m2w[UNITW ? 2 : 2 * NP * Q];
    if (!UNITW) {
        for (int i = (int)threadIdx.x; i < 2 * NP * Q; i += (int)blockDim.x) {
            lds_w[i] = (p.gw != nullptr && i < p.D) ? p.gw[i] : 1.0;
            lds_mu[i] = (p.gmu != nullptr && i < p.D) ? p.gmu[i] : 0.0;
            lds_m2w[i] = -2.0 * lds_w[i];
        }
    }
    auto_begin();
    kd_tables_to_lds();          // (ends with the workgroup barrier)
    if (p.clock_probe != nullptr && blockIdx.x == (gridDim.x >> 1) && threadIdx.x == 0) {
        p.clock_probe[2] = __builtin_amdgcn_s_memtime(); p.clock_probe[3] = __builtin_amdgcn_s_memrealtime();
    }
    const int D = p.D;
    const PairCtx<NP, Q> cx = make_pctx<NP, Q>(D);
    const int nsteps = ONESTEP ? 1 : kl.nsteps;
    guchar* const accept_out = p.accept != nullptr ? p.accept + kl.t0 * (unsigned long long)p.nchains : nullptr;

    double sig[E];                       // (only MH has proposal scales)
    const auto wv = [&](int e) { return UNITW ? 1.0 : lds_w[2 * ((e >> 1) * Q + cx.q) + (e & 1)]; };
    const auto mv = [&](int e) { return UNITW ? 0.0 : lds_mu[2 * ((e >> 1) * Q + cx.q) + (e & 1)]; };
    const auto m2wv = [&](int e) { return UNITW ? -2.0 : lds_m2w[2 * ((e >> 1) * Q + cx.q) + (e & 1)]; };
    // inside the leapfrog loop the reads are volatile: otherwise they are hoisted out of the loop into 4*NP registers
    const auto wvl = [&](int e) { return UNITW ? 1.0 : *(volatile const double*)&lds_w[2 * ((e >> 1) * Q + cx.q) + (e & 1)]; };
    const auto mvl = [&](int e) { return UNITW ? 0.0 : *(volatile const double*)&lds_mu[2 * ((e >> 1) * Q + cx.q) + (e & 1)]; };
    const auto m2wvl = [&](int e) { return UNITW ? -2.0 : *(volatile const double*)&lds_m2w[2 * ((e >> 1) * Q + cx.q) + (e & 1)]; };
    if (SAMPLER == KLARA_SAMPLER_MH) load_pair_param<NP, Q>(cx, p.vecparam, D, 1.0, sig);   // proposal scales (the slice sampler reads its widths where it uses them)
    const double gconst = USERPAIR ? 0.0 : p.gconst;

    // accept draw: slot S = ceil(D/2) = D/2.  (NP-1)*Q < D/2 <= NP*Q, so when the layout has padding (D/2 < NP*Q) the
    // slot is the last pair of lane S % Q and its Box-Muller already formed u and log(u).
    const int acc_slot = (D + 1) >> 1;
    const bool acc_free = acc_slot < NP * Q && !PairShare<Q>::second(NP - 1);
    const int acc_lane = (cx.lane - cx.q) + (acc_slot & (Q - 1));

    const long long nwaves = (long long)gridDim.x * (blockDim.x >> 6);
    const long long wave0 = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    // one chain group per wavefront (the launcher sizes the grid for it); a persistent loop here makes the compiler
    // park the polynomial constants in VGPRs for the whole kernel
    unsigned wave_acc = 0;                                  // accepted proposals of this wavefront's chains (KAuto)
    KLARA_DT_GROUP_LOOP {
        const long long first_chain = grp * CPW;
        const long long left = p.nchains - first_chain;
        const int here = left < CPW ? (int)left : CPW;
        const bool chain_ok = cx.cw < here;
        const long long chain = first_chain + cx.cw;
        const unsigned long long gchain = (unsigned long long)(p.chain_offset + chain);
        const __amdgpu_buffer_rsrc_t wx = group_window(p.X, first_chain, here, D);
        const __amdgpu_buffer_rsrc_t wg = group_window(p.GR, first_chain, here, D);

        double x[E];
        load_pairs<NP, Q>(cx, wx, x);
        // The gradient of this target family is a function of the value alone and GR always holds gradlogtarget(X) (set by
        // initialize! and by every accepted transition).  It is therefore neither loaded nor kept: wherever iterate! reads
        // the current gradient it is re-formed from x — the same operations that produced the stored bits — and the
        // proposal's gradient is formed from the proposal when it is written out.  Half the state read, and 8*NP fewer
        // registers over the launch (x, proposal, normals and running sums are what a lane holds).
        // user pair closure at the lane's pair pi: -term and the two partial derivatives; padding pairs / the missing half of an odd
        // D's last pair contribute exact zeros (their values, normals and derivatives stay 0 through every update)
        const auto user_pair = [&](int pi, double a, double b, double& nt, double& g0, double& g1) {
            // (the closure is CALLED for real pairs only, pair < ceil(D/2): it may index its data block by pair or by coordinate —
            // the oracle's ko_pair_eval makes exactly these calls.  Only a lane's last pair can be padding.)
            double u0 = 0.0, u1 = 0.0, u = 0.0;
            const bool ok0 = pi < NP - 1 || cx.last_ok, ok1 = pi < NP - 1 || cx.last_full;
            if (ok0) u = KLARA_PAIR_CALL(a, b, pi * Q + cx.q, &u0, &u1);
            nt = ok0 ? -u : 0.0; g0 = ok0 ? u0 : 0.0; g1 = ok1 ? u1 : 0.0;
        };
        const auto grad_of = [&](const double (&v)[E], double (&out)[E]) {
            if constexpr (USERPAIR) {
#pragma unroll
                for (int pi = 0; pi < NP; ++pi) { double nt; user_pair(pi, v[2 * pi], v[2 * pi + 1], nt, out[2 * pi], out[2 * pi + 1]); }
            } else {
#pragma unroll
                for (int e = 0; e < E; ++e) { double term_; diag_elem<UNITW>(v[e], wv(e), m2wv(e), mv(e), term_, out[e]); }
            }
        };
        double lt = p.LT[chain_ok ? chain : 0];
        unsigned long long nacc = 0;
        bool stuck = false;                                // slice sampler: step-out / shrink ran out of attempts
        // saved-sample monitors (MON).  Running sums are kept in sojourn form (KParams::held): the save rule only counts the saved
        // steps a chain spends at its current state, and the state is folded into sum / sumsq (held * x, held * x^2) when the
        // chain leaves it.  A chain's sums are loaded the first time that happens in the launch and written back only then, so
        // a chain that does not move (98-99.6 % of the transitions of the drift-0.9 job) costs no running-sum traffic at all:
        // 210 MB per launch of 65,536 x 100 otherwise, 40-90 us of every launch.
        const bool do_sum = MON && p.sum != nullptr;
        double sm[MEMSUM ? 2 : E], sq[MEMSUM ? 2 : E];
        bool sums_loaded = false;                          // (per chain: uniform over the chain's Q lanes)
        long long held = 0;
        __amdgpu_buffer_rsrc_t wsum = wx, wsq = wx;
        if (do_sum) {
#pragma unroll
            for (int e = 0; e < (MEMSUM ? 2 : E); ++e) { sm[e] = 0.0; sq[e] = 0.0; }
            wsum = group_window(p.sum, first_chain, here, D); wsq = group_window(p.sumsq, first_chain, here, D);
            held = p.held[chain_ok ? chain : 0];
        }
        int sphase = kl.save_phase0;
        long long scol = kl.save_col0;
        // tuner state (tuners.jl:5-10): per chain, or one pooled entry per GPU
        const bool per_chain_tune = KCNT && !KPOOLED;
        const long long c0 = chain_ok ? chain : 0;
        TuneRegs tn;
        if (per_chain_tune) tn = { p.tune_step[c0], p.tune_accepted[c0], p.tune_proposed[c0], p.tune_totproposed[c0], 0, 0.0, 0.0 };
        else if (KPOOLED) tn = { p.tune_step[0], p.tune_accepted[0], 0, 0, 0, 0.0, 0.0 };
        else tn = { DA ? p.tune_step[c0] : p.step0, 0, 0, 0, 0, 0.0, 0.0 };
        if (DA) { tn.epsbar = p.da_epsbar[c0]; tn.hbar = p.da_hbar[c0]; }
        const long long acc0 = tn.accepted;
        tn.phase = per_chain_tune ? (int)(tn.proposed % p.period) : 0;

        for (int s = 0; s < nsteps; ++s) {
            const unsigned long long t = kl.t0 + (unsigned long long)s;
            if (KCNT) tune_count_proposal(p, tn);
            double xp[E];
            const auto put_xp = [&](int pi, double a, double b) { xp[2 * pi] = a; xp[2 * pi + 1] = b; };
            // Q = 4 sums in the order of the 8-lane layout (the order klara_get_layout reports): lane q of 4 holds the pairs of the
            // 8-lane layout's lanes q (its even pairs p) and q + 4 (odd p), each in ascending order; two partial sums per lane, the
            // strides 1 and 2 of the butterfly on both, stride 4 is their sum — bit for bit what 8 lanes per chain produce.
            constexpr int NR = Q == 4 ? 2 : 1;
            double red[3 * NR] = {}, red1[1], red2[2];
            double u_last = 0.5, lg_last = 0.0;
            uint32_t nstash[4] = { 0u, 0u, 0u, 0u };             // words (z, w) of the blocks whose second pair is still to come (pair_normals)
            bool acc;
            double ltp, a_da = 0.0;

            // One transition per launch: all proposal normals are drawn first — they do not depend on the chain state, so the
            // Philox/Box-Muller work (~4000 issue cycles) runs while the state loads issued above are still in flight.  Fused
            // launches draw a pair where it is consumed instead (no array of 2*NP normals: 4*NP registers less).
            // (on the 4-lane form, where 4*NP registers decide the occupancy; with 8 or more lanes per chain the array costs nothing
            // and drawing first measured faster: 15.3 vs 17.1 us per transition for the monitored MALA job)
            constexpr bool ZFIRST = (ONESTEP || Q != 4) && SAMPLER != KLARA_SAMPLER_HMC && !SLICE;
            double z[ZFIRST ? E : 2];
            if (ZFIRST) {
#pragma unroll
                for (int pi = 0; pi < NP; ++pi) {
                    pair_normals<NP, Q>(cx, p.seed, gchain, t, pi, nstash, z[2 * pi], z[2 * pi + 1], u_last, lg_last);
                    KLARA_DT_PAIR_FENCE(pi);
                }
            }
            if constexpr (SLICE) {
                if (do_sum && __any(held > 0))                                     // the slice sampler always moves: fold first
                    diagt_fold<NP, Q>(cx, wsum, wsq, held > 0, sums_loaded, held, x, sm, sq);
            }
            if (SLICE) {                                                           // iterate/SliceSampler.jl:60-109
                // The target is a sum of per-coordinate terms, lt = c - sum_i w_i (x_i - mu_i)^2, and a coordinate update only ever compares the
                // log-target of a candidate with the slice level log(rand()) + lt (:66, :76, :84, :95): with t_i the moving coordinate's term,
                //     lt(candidate) > log(rand()) + lt   <=>   t_i(current) - t_i(candidate) > log(rand())
                // — the other coordinates cancel.  So the updates of a transition do not depend on each other, and instead of the chain's Q lanes
                // walking through the D coordinates together (every probe a full evaluation: the lane's terms re-added and a Q-lane reduction,
                // 7 of 8 lanes repeating the scalar work), every lane updates ITS OWN coordinates, all 64 lanes of the wavefront at once: no
                // cross-lane traffic until the new state's log-target is formed at the end (one full evaluation in the layout's order).  The
                // comparison in this difference form is a deliberate deviation from the literal arithmetic, shared with the
                // oracle for this layout; loops run until every lane of the wavefront is done (__any).
                // (the loops are short dependent chains between wave-wide votes: the kernel lives on wavefronts per SIMD, so nothing but the value is kept
                // in registers over the transition — a coordinate's current term is re-formed where its update starts, its width read where it is used)
                acc = true;
                const __amdgpu_buffer_rsrc_t wwid = __builtin_amdgcn_make_buffer_rsrc((void*)p.vecparam, 0, D * 8, 0x00020000);
#pragma unroll
                for (int e = 0; e < E; ++e) {                                                  // :65 (the lane's coordinates in ascending order)
                    const int i = 2 * ((e >> 1) * Q + cx.q) + (e & 1);
                    const bool live = chain_ok && i < D;
                    const uint32_t base = (uint32_t)(live ? i : 0) << KLARA_SLICE_ATT_BITS;
                    const kd_u32x4 b0 = kd_stream_block(p.seed, gchain, t, base);
                    const double lgu = kd_log_u01(kd_uniform_xy(b0));                          // :66 log(rand()); the slice level is lgu + lt
                    const double ru = kd_uniform_zw(b0);                                       // :71
                    const double wi_t = UNITW ? 1.0 : wv(e), mi_t = UNITW ? 0.0 : mv(e);
                    const double xi = x[e], wd = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(wwid, (unsigned)(live ? i : 0) * 8u, 0, 0));
                    // (round 6) a pair closure: the target is a sum of terms of ONE pair each, so an update of coordinate i = 2P (+ 1) compares the term of pair P
                    // with the candidate in place of x_i against the pair's current term — the other pairs cancel like the other coordinates of the diagonal
                    // target.  A lane holds whole pairs and walks its elements in ascending order: coordinate 2P + 1 sees the new x_2P (:65, :108).
                    double tcur, gd_;
                    if constexpr (USERPAIR) { double g1_; user_pair(e >> 1, x[e & ~1], x[e | 1], tcur, gd_, g1_); }
                    else diag_elem<UNITW>(xi, wi_t, -2.0 * wi_t, mi_t, tcur, gd_);
                    double Li = xi - ru * wd;                                                  // :72
                    double Ri = xi + (1.0 - ru) * wd;                                          // :73
                    const auto term_of = [&](double cand) -> double {
                        double tc, gd;
                        if constexpr (USERPAIR) { double g1; user_pair(e >> 1, (e & 1) ? x[e & ~1] : cand, (e & 1) ? cand : x[e | 1], tc, gd, g1); }
                        else diag_elem<UNITW>(cand, wi_t, -2.0 * wi_t, mi_t, tc, gd);
                        return tc;
                    };
                    if (p.stepout) {                                                           // :75-89
                        double dl = tcur - term_of(Li);
                        int guard = 0;
                        while (true) {
                            bool go = live && !stuck && (dl > lgu);
                            if (go && ++guard > KLARA_SLICE_MAX_ATT) { stuck = true; go = false; }
                            if (!__any(go)) break;
                            const double Ln = Li - wd;
                            const double dn = tcur - term_of(go ? Ln : Li);
                            if (go) { Li = Ln; dl = dn; }
                        }
                        double dr = tcur - term_of(Ri);
                        guard = 0;
                        while (true) {
                            bool go = live && !stuck && (dr > lgu);
                            if (go && ++guard > KLARA_SLICE_MAX_ATT) { stuck = true; go = false; }
                            if (!__any(go)) break;
                            const double Rn = Ri + wd;
                            const double dn = tcur - term_of(go ? Rn : Ri);
                            if (go) { Ri = Rn; dr = dn; }
                        }
                    }
                    double xprime = xi;
                    bool done = !live || stuck;
                    kd_u32x4 ab = b0;                                                          // the block of the current pair of attempts
                    for (uint32_t a = 1;; ++a) {                                               // :91-106
                        if (!done && a > KLARA_SLICE_MAX_ATT) { stuck = true; done = true; }
                        if (!__any(!done)) break;
                        if (a & 1u) ab = kd_stream_block(p.seed, gchain, t, kd_slice_attempt_slot(base, a));   // (a is wave-uniform: attempts 2k - 1 and 2k share a block)
                        const double u = (a & 1u) ? kd_uniform_xy(ab) : kd_uniform_zw(ab);
                        const double cand = u * (Ri - Li) + Li;                                // :92-93
                        const double tc = term_of(done ? xprime : cand);                       // :94
                        if (!done) {
                            xprime = cand;
                            if (tcur - tc > lgu) done = true;                                  // :95
                            else if (cand > xi) Ri = cand;                                     // :98
                            else if (cand < xi) Li = cand;                                     // :100
                            else { stuck = true; done = true; }                                // :102
                        }
                    }
                    if (live && !stuck) x[e] = xprime;                                         // :108
                }
                red1[0] = 0.0;
                if constexpr (USERPAIR) {
#pragma unroll
                    for (int pi = 0; pi < NP; ++pi) { double nt, g0_, g1_; user_pair(pi, x[2 * pi], x[2 * pi + 1], nt, g0_, g1_); red1[0] = red1[0] + nt; }
                } else {
#pragma unroll
                for (int e = 0; e < E; ++e) { double te, gd; diag_elem<UNITW>(x[e], wv(e), m2wv(e), mv(e), te, gd); red1[0] = red1[0] + te; }
                }
                group_allreduce<1>(red1, Q, cx.lane);
                lt = gconst - red1[0];                                                         // the new state's log-target: one full evaluation
                ltp = lt;
#pragma unroll
                for (int e = 0; e < E; ++e) xp[e] = x[e];          // (the commit below is then a no-op)
            } else if (SAMPLER == KLARA_SAMPLER_MH) {                              // iterate/MH.jl:72-124
                const auto mh_elem = [&](int e, double ze, int r) -> double {
                    const double xe = x[e] + sig[e] * ze;                                      // MH.jl:79
                    double term, gd;
                    diag_elem<UNITW>(xe, wv(e), m2wv(e), mv(e), term, gd);          // :81
                    red[r] = red[r] + term;
                    return xe;
                };
#pragma unroll
                for (int pi = 0; pi < NP; ++pi) {
                    double z0 = ZFIRST ? z[(2 * pi) % (ZFIRST ? E : 2)] : 0.0, z1 = ZFIRST ? z[(2 * pi + 1) % (ZFIRST ? E : 2)] : 0.0;
                    if (!ZFIRST) pair_normals<NP, Q>(cx, p.seed, gchain, t, pi, nstash, z0, z1, u_last, lg_last);
                    const int r = (NR == 2 && (pi & 1)) ? 3 : 0;
                    if constexpr (USERPAIR) {
                        const double a = x[2 * pi] + sig[2 * pi] * z0, b = x[2 * pi + 1] + sig[2 * pi + 1] * z1;      // MH.jl:79
                        double nt, g0, g1;
                        user_pair(pi, a, b, nt, g0, g1);                                                           // :81
                        red[r] = red[r] + nt;
                        put_xp(pi, a, b);
                    } else {
                        const double a = mh_elem(2 * pi, z0, r), b = mh_elem(2 * pi + 1, z1, r);
                        put_xp(pi, a, b);
                    }
                    if (!ZFIRST) KLARA_DT_PAIR_FENCE(pi);
                }
                if (NR == 2) {
                    red2[0] = red[0]; red2[1] = red[3];
                    group_allreduce<2>(red2, Q, cx.lane);
                    red1[0] = red2[0] + red2[1];
                } else {
                    red1[0] = red[0];
                    group_allreduce<1>(red1, Q, cx.lane);
                }
                ltp = gconst - red1[0];
                const double ratio = ltp - lt;                                                 // :83
                acc = ratio > 0.0;                                                             // :97
                if (acc_free) acc = acc || ratio > lane_bcast(lg_last, acc_lane);
                else if (!acc && ratio > KD_LOG_UMIN_GUARD)
                    acc = ratio > kd_log_u01(kd_accept_uniform(kd_stream_block(p.seed, gchain, t, (uint32_t)acc_slot)));
            } else if (SAMPLER == KLARA_SAMPLER_MALA) {                            // iterate/MALA.jl:78-128
                const double h_ = tn.step, halfh = 0.5 * h_, sq = KCNT ? __builtin_sqrt(h_) : p.sqrt_step0;
                const double half_inv_h = 0.5 * (KCNT ? 1.0 / h_ : p.inv_step0);
                const auto mala_elem = [&](int e, double ze, int r) -> double {
                    double term, ge, gpe;
                    diag_elem<UNITW>(x[e], wv(e), m2wv(e), mv(e), term, ge);        // (the current gradient, re-formed)
                    const double m_ = x[e] + halfh * ge;                                       // :83
                    const double xe = m_ + sq * ze;                                            // :84
                    diag_elem<UNITW>(xe, wv(e), m2wv(e), mv(e), term, gpe);         // :86
                    red[r] = red[r] + term;
                    const double q1 = m_ - xe;
                    red[r + 1] = red[r + 1] + (q1 * q1) * half_inv_h;                          // :90
                    const double mup = xe + halfh * gpe;                                       // :91
                    const double q2 = mup - x[e];
                    red[r + 2] = red[r + 2] + (q2 * q2) * half_inv_h;                          // :92
                    return xe;
                };
#pragma unroll
                for (int pi = 0; pi < NP; ++pi) {
                    double z0 = ZFIRST ? z[(2 * pi) % (ZFIRST ? E : 2)] : 0.0, z1 = ZFIRST ? z[(2 * pi + 1) % (ZFIRST ? E : 2)] : 0.0;
                    if (!ZFIRST) pair_normals<NP, Q>(cx, p.seed, gchain, t, pi, nstash, z0, z1, u_last, lg_last);
                    const int r = (NR == 2 && (pi & 1)) ? 3 : 0;
                    if constexpr (USERPAIR) {
                        double nt, ge0, ge1, gp0, gp1;
                        user_pair(pi, x[2 * pi], x[2 * pi + 1], nt, ge0, ge1);                    // (the current gradient, re-formed)
                        const double m0 = x[2 * pi] + halfh * ge0, m1 = x[2 * pi + 1] + halfh * ge1;               // :83
                        const double a = m0 + sq * z0, b = m1 + sq * z1;                                           // :84
                        user_pair(pi, a, b, nt, gp0, gp1);                                                         // :86
                        red[r] = red[r] + nt;
                        const double q10 = m0 - a, q11 = m1 - b;
                        red[r + 1] = red[r + 1] + (q10 * q10) * half_inv_h;                                        // :90
                        red[r + 1] = red[r + 1] + (q11 * q11) * half_inv_h;
                        const double q20 = (a + halfh * gp0) - x[2 * pi], q21 = (b + halfh * gp1) - x[2 * pi + 1]; // :91
                        red[r + 2] = red[r + 2] + (q20 * q20) * half_inv_h;                                        // :92
                        red[r + 2] = red[r + 2] + (q21 * q21) * half_inv_h;
                        put_xp(pi, a, b);
                    } else {
                        const double a = mala_elem(2 * pi, z0, r), b = mala_elem(2 * pi + 1, z1, r);
                        put_xp(pi, a, b);
                    }
                    if (!ZFIRST) KLARA_DT_PAIR_FENCE(pi);
                }
                group_allreduce<3 * NR>(red, Q, cx.lane);
                if (NR == 2) { red[0] = red[0] + red[3]; red[1] = red[1] + red[4]; red[2] = red[2] + red[5]; }
                ltp = gconst - red[0];
                double ratio = ltp - lt;                                                       // :88
                ratio += red[1];                                                               // :90
                ratio -= red[2];                                                               // :92
                acc = ratio > 0.0;                                                             // :94
                if (acc_free) acc = acc || ratio > lane_bcast(lg_last, acc_lane);
                else if (!acc && ratio > KD_LOG_UMIN_GUARD)
                    acc = ratio > kd_log_u01(kd_accept_uniform(kd_stream_block(p.seed, gchain, t, (uint32_t)acc_slot)));
            } else {                                                               // iterate/HMC.jl:124-201
                const double eps = tn.step, halfe = 0.5 * eps;
                double mom[E], gp[E];
                // (Q = 4: two partial sums per reduction — the pairs of the 8-lane layout's lanes q (even pi) and q + 4 (odd pi) — as in the MH / MALA branches:
                // strides 1 and 2 of the butterfly on both, stride 4 is their sum: the bits 8 lanes per chain produce)
                double k0[NR] = {};
#pragma unroll
                for (int pi = 0; pi < NP; ++pi) {
                    pair_normals<NP, Q>(cx, p.seed, gchain, t, pi, nstash, mom[2 * pi], mom[2 * pi + 1], u_last, lg_last);   // :135
                    const int r = (NR == 2 && (pi & 1)) ? 1 : 0;
                    k0[r] = k0[r] + mom[2 * pi] * mom[2 * pi];
                    k0[r] = k0[r] + mom[2 * pi + 1] * mom[2 * pi + 1];
                }
                group_allreduce<NR>(k0, Q, cx.lane);
                if (NR == 2) k0[0] = k0[0] + k0[1];
                const double H0 = lt - 0.5 * k0[0];                                            // :137
#pragma unroll
                for (int e = 0; e < E; ++e) xp[e] = x[e];                                      // :139
                grad_of(x, gp);                                                                // :140 (re-formed)
                const int nl = DA ? (chain_ok ? da_nleaps(p, eps) : 1) : p.nleaps;             // :142-144 (padding lanes: 1)
                // leapfrog! nl times (:146-155, samplers.jl:122-134) in its merged form (deliberate deviation (7),
                // mirrored by the oracle): adjacent half-kicks are one update, every update is one fma
#pragma unroll
                for (int e = 0; e < E; ++e) mom[e] = kd_fma(halfe, gp[e], mom[e]);
                const int nlmax = DA ? wave_max_int(nl) : nl;
                for (int l = 0; l < nlmax; ++l) {
                    const bool go = !DA || l < nl;
                    const double kf = l + 1 < nl ? eps : halfe;
                    if constexpr (USERPAIR) {
#pragma unroll
                        for (int pi = 0; pi < NP; ++pi) {
                            const double xa = kd_fma(eps, mom[2 * pi], xp[2 * pi]), xb = kd_fma(eps, mom[2 * pi + 1], xp[2 * pi + 1]);
                            xp[2 * pi] = go ? xa : xp[2 * pi]; xp[2 * pi + 1] = go ? xb : xp[2 * pi + 1];
                            double nt, ga, gb;
                            user_pair(pi, xp[2 * pi], xp[2 * pi + 1], nt, ga, gb);
                            gp[2 * pi] = go ? ga : gp[2 * pi]; gp[2 * pi + 1] = go ? gb : gp[2 * pi + 1];
                            const double na = kd_fma(kf, gp[2 * pi], mom[2 * pi]), nb = kd_fma(kf, gp[2 * pi + 1], mom[2 * pi + 1]);
                            mom[2 * pi] = go ? na : mom[2 * pi]; mom[2 * pi + 1] = go ? nb : mom[2 * pi + 1];
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < E; ++e) {
                            const double x1 = kd_fma(eps, mom[e], xp[e]);
                            xp[e] = go ? x1 : xp[e];
                            double term, g1;
                            diag_elem<UNITW>(xp[e], 1.0, m2wvl(e), mvl(e), term, g1);       // (term unused in the leapfrog)
                            gp[e] = go ? g1 : gp[e];
                            const double m2 = kd_fma(kf, gp[e], mom[e]);
                            mom[e] = go ? m2 : mom[e];
                        }
                    }
                }
                if constexpr (USERPAIR) {
#pragma unroll
                    for (int pi = 0; pi < NP; ++pi) {
                        double nt, ga, gb;
                        user_pair(pi, xp[2 * pi], xp[2 * pi + 1], nt, ga, gb);                     // :157
                        red[0] = red[0] + nt;
                        red[1] = red[1] + mom[2 * pi] * mom[2 * pi];
                        red[1] = red[1] + mom[2 * pi + 1] * mom[2 * pi + 1];
                    }
                } else {
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    double term, gd;
                    diag_elem<UNITW>(xp[e], wv(e), m2wv(e), mv(e), term, gd);   // :157
                    const int r = (NR == 2 && ((e >> 1) & 1)) ? 3 : 0;          // (Q = 4: the odd pairs' partials, see the opening sum)
                    red[r] = red[r] + term;
                    red[r + 1] = red[r + 1] + mom[e] * mom[e];
                }
                }
                if (NR == 2) {
                    double red4[4] = { red[0], red[1], red[3], red[4] };
                    group_allreduce<4>(red4, Q, cx.lane);
                    red2[0] = red4[0] + red4[2]; red2[1] = red4[1] + red4[3];
                } else {
                    red2[0] = red[0]; red2[1] = red[1];
                    group_allreduce<2>(red2, Q, cx.lane);
                }
                ltp = gconst - red2[0];
                const double H1 = ltp - 0.5 * red2[1];                                         // :159
                const double ratio = H1 - H0;                                                  // :161
                const double ex = kd_exp(ratio);
                const double a = 1.0 < ex ? 1.0 : ex;                                          // :163
                a_da = a;
                const double u = acc_free ? lane_bcast(u_last, acc_lane)
                                          : kd_accept_uniform(kd_stream_block(p.seed, gchain, t, (uint32_t)acc_slot));
                acc = u < a;                                                                   // :165
            }

            if (accept_out != nullptr && chain_ok && cx.q == 0) accept_out[(long long)s * p.nchains + chain] = acc ? 1 : 0;
            if (KCNT && acc && !SLICE) tn.accepted += 1;                // (the slice sampler never counts accepts)
            if (DA) da_update(p, tn, (long long)t + 1, a_da);           // iterate/HMC.jl:225-249
            if (per_chain_tune && !DA) tuning_block(p, tn);             // iterate/MALA.jl:130-152, HMC.jl:203-224
            else if (DA && per_chain_tune && tn.phase == 0 && (long long)t + 1 <= p.da_nadapt) {   // verbose report block
                tn.totproposed += tn.proposed; tn.accepted = 0; tn.proposed = 0;
            }
            if constexpr (ONESTEP) {
                wave_acc += (acc && chain_ok && cx.q == 0) ? 1u : 0u;
                if (acc) {                               // accepted proposal: registers -> HBM, nothing else moves
                    store_pairs<NP, Q>(cx, wx, xp);
                    if (NEEDG) { double gq[E]; grad_of(xp, gq); store_pairs<NP, Q>(cx, wg, gq); }
                    if (chain_ok && cx.q == 0) { p.LT[chain] = ltp; p.naccept[chain] += 1ull; }
                }
            } else {
                nacc += acc ? 1ull : 0ull;
                if (!SLICE && do_sum && __any(chain_ok && acc && held > 0)) {          // a chain of this wavefront leaves its state
                    const bool fold = chain_ok && acc && held > 0;
#ifdef KLARA_DT_FOLD_RMW
                    if constexpr (MEMSUM) diagt_fold_rmw<NP, Q>(cx, wsum, wsq, fold, held, x);
#else
                    // (both forms in one kernel, chosen by how many chains of the wavefront move, cost more than they save: 84 B of scratch and
                    // 14.2 against 13.7 / 13.55 us per transition — profiles/r4_ab_fold_atomic.txt)
                    if constexpr (MEMSUM) diagt_fold_atomic<NP, Q>(cx, p.sum + first_chain * D, p.sumsq + first_chain * D, fold, held, x);
#endif
                    else diagt_fold<NP, Q>(cx, wsum, wsq, fold, sums_loaded, held, x, sm, sq);
                }
                if (acc) {
#pragma unroll
                    for (int e = 0; e < E; ++e) x[e] = xp[e];
                    lt = ltp;
                }
            }
            // save rule: BasicMCJob.jl:226-231 with postrange = (burnin+1):thinning:nsteps (BasicMCRange.jl:36); the
            // phase / column bookkeeping comes from the host (klara_run_async), as in the group-layout kernel
            const long long i1 = (long long)t + 1;
            if (MON && i1 > p.burnin && i1 <= p.nsteps_total) {
                if (sphase == 0) {
                    if (do_sum) held += 1;
                    if (scol < p.hist_cols) {
                        const long long col0 = scol * p.nchains + first_chain;
                        if (p.hist != nullptr) store_pairs<NP, Q>(cx, group_window(p.hist, col0, here, D), x);
                        if (NEEDG && p.hist_g != nullptr) { double gq[E]; grad_of(x, gq); store_pairs<NP, Q>(cx, group_window(p.hist_g, col0, here, D), gq); }
                        if (p.hist_lt != nullptr && chain_ok && cx.q == 0) p.hist_lt[scol * p.nchains + chain] = lt;
                    }
                    ++scol;
                }
                sphase = (sphase + 1 == (int)p.thinning) ? 0 : sphase + 1;
            }
        }
        if (do_sum) {
            if constexpr (!MEMSUM) {
                if (__any(sums_loaded)) {                  // only the chains that left a state during the launch write their sums back
                    store_pairs_if<NP, Q>(cx, wsum, sums_loaded, sm);
                    store_pairs_if<NP, Q>(cx, wsq, sums_loaded, sq);
                }
            }
            if (chain_ok && cx.q == 0) p.held[chain] = held;
        }
        // (a per-lane flag: the lane whose coordinate ran out of attempts stops, the chain's other lanes finish the transition — the oracle returns at the
        // first stuck coordinate, so after KLARA_ERR_SLICE_STUCK the device state is NOT the oracle's: unspecified, include/klara_hip.h)
        if (SLICE && stuck && chain_ok) klara_raise(p.error_flag, KLARA_ERR_SLICE_STUCK);
        if (!ONESTEP) wave_acc += (chain_ok && cx.q == 0) ? (unsigned)nacc : 0u;     // (per-lane partial; summed in auto_finish)
        if (!ONESTEP && nacc != 0) {
            store_pairs<NP, Q>(cx, wx, x);
            if (NEEDG) { double gq[E]; grad_of(x, gq); store_pairs<NP, Q>(cx, wg, gq); }
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
    if (p.clock_probe != nullptr && blockIdx.x == (gridDim.x >> 1) && threadIdx.x == 0) {
        p.clock_probe[0] = __builtin_amdgcn_s_memtime(); p.clock_probe[1] = __builtin_amdgcn_s_memrealtime();
    }
    auto_finish(ka, wave_acc);
}

// initialize!(pstate, parameter, sampler) for layout kind 3: lt (and the gradient) at X, finiteness check
template <int NP, int Q, bool USERPAIR = false>
__global__ __launch_bounds__(256) void k_diagt_init(const KParams p, int needgrad)
{
#ifndef KLARA_USER_PAIR_TARGET
    static_assert(!USERPAIR, "pair closures exist in run-time compiled translation units only");
#endif
    constexpr int E = 2 * NP, CPW = 64 / Q;
    const int D = p.D;
    const PairCtx<NP, Q> cx = make_pctx<NP, Q>(D);
    const long long grp = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));   // wave-uniform: scalar windows
    const long long first_chain = grp * CPW;
    const long long left = p.nchains - first_chain;
    const int here = left < CPW ? (left > 0 ? (int)left : 0) : CPW;
    const bool chain_ok = cx.cw < here;
    const long long chain = first_chain + cx.cw;
    double w[E], mu[E], x[E], g[E], red[1] = { 0.0 };
    load_pair_param<NP, Q>(cx, p.gw, D, 1.0, w);
    load_pair_param<NP, Q>(cx, p.gmu, D, 0.0, mu);
    load_pairs<NP, Q>(cx, group_window(p.X, first_chain, here, D), x);
    bool bad = false;
    if constexpr (USERPAIR) {
#pragma unroll
        for (int pi = 0; pi < NP; ++pi) {
            double u0 = 0.0, u1 = 0.0, u = 0.0;
            const bool ok0 = pi < NP - 1 || cx.last_ok, ok1 = pi < NP - 1 || cx.last_full;
            if (ok0) u = KLARA_PAIR_CALL(x[2 * pi], x[2 * pi + 1], pi * Q + cx.q, &u0, &u1);       // real pairs only (see k_diagt)
            red[0] = red[0] + (ok0 ? -u : 0.0);
            g[2 * pi] = ok0 ? u0 : 0.0; g[2 * pi + 1] = ok1 ? u1 : 0.0;
            bad = bad || !kfinite(g[2 * pi]) || !kfinite(g[2 * pi + 1]);
        }
    } else {
#pragma unroll
    for (int e = 0; e < E; ++e) {
        double term;
        diag_elem<false>(x[e], w[e], -2.0 * w[e], mu[e], term, g[e]);
        red[0] = red[0] + term;
        bad = bad || !kfinite(g[e]);
    }
    }
    group_allreduce<1>(red, Q, cx.lane);
    const double lt = (USERPAIR ? 0.0 : p.gconst) - red[0];
    bad = chain_ok && ((needgrad && bad) || !kfinite(lt));
    if (needgrad) store_pairs<NP, Q>(cx, group_window(p.GR, first_chain, here, D), g);
    if (chain_ok && cx.q == 0) p.LT[chain] = lt;
    if (bad) klara_raise(p.error_flag, KLARA_ERR_NONFINITE_INIT);
}
