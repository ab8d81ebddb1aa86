// klara_kernels.h — gfx950 device code for the many-chain transition kernels ("group layout").
//
// Layout (kind 0).  A chain's D-vector is spread over G = 2^k lanes of a wavefront,
// E contiguous elements per lane (element i lives on lane i / E of its group); a 64-lane wavefront
// carries 64 / G chains.  D = 100 -> E = 4, G = 32: two chains per wavefront, lanes 0..24 of each half hold
// (4q..4q+3), so two Philox4x32-10 blocks per lane yield exactly that lane's four proposal normals (E = 2, G = 64 is
// the one-chain-per-wavefront variant; klara_api.hip select_layout picks E).
// D = 4 (swiss logistic regression) -> E = 4, G = 1: one chain per lane, no cross-lane traffic at all.
// State matrices are (nchains x D) row-major in HBM: a wavefront reads/writes contiguous rows.
//
// Reductions over a chain (dot, sum) are: lane partial over its E elements (ascending), then an xor
// butterfly over the G lanes — DPP quad_perm / row_half_mirror / row_mirror for strides 1,2,4,8 and
// ds_bpermute for 16,32.  The CPU oracle sums in the same order (oracle/klara_oracle.c ko_reduce).
//
// The transition arithmetic restates src/samplers/iterate/{MH,MALA,HMC,SliceSampler}.jl expression by
// expression (no fma contraction: build with -ffp-contract=off); citations are on each step.
#pragma once
#ifdef __HIPCC_RTC__                 // run-time compilation of a user-defined target (klara_custom.h): flat header names
#include "detmath.h"
#include "klara_hip.h"
#else
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "detmath.h"
#include "../../include/klara_hip.h"
#endif

#ifndef KLARA_E2_DIAG_WAVES
#define KLARA_E2_DIAG_WAVES 3   // min waves per SIMD requested for the 2-elements-per-lane kernels on the diagonal Gaussian (cfg 1 as replicas)
#endif
#ifndef KLARA_E4_WAVES
#define KLARA_E4_WAVES 2   // min waves per SIMD requested for the E=4 kernels (register budget 256)
#endif
// data rows of the logistic target that go through the stages of an evaluation together (LogisticTarget::eval)
#ifndef KLARA_LOGIT_BATCH
#define KLARA_LOGIT_BATCH 4
#endif
#define KLARA_LOGIT_BATCH_OF(E) ((E) <= 4 ? KLARA_LOGIT_BATCH : ((KLARA_LOGIT_BATCH) > 3 ? 3 : KLARA_LOGIT_BATCH))
template <int N> struct KInt { static constexpr int value = N; };
template <int N> __device__ __forceinline__ constexpr int kstride(KInt<N>) { return N; }
__device__ __forceinline__ constexpr int kstride(int v) { return v; }
// nothing is scheduled across a stage boundary of the batched row evaluation
// (KLARA_PIN(v): the value is "produced" here as far as the compiler knows, so arithmetic on it cannot be hoisted above this point)
#if defined(__HIP_DEVICE_COMPILE__)
#define KLARA_SCHED_STAGE() __builtin_amdgcn_sched_barrier(0)
#define KLARA_PIN(v) asm volatile("" : "+v"(v))
#else
#define KLARA_SCHED_STAGE() ((void)0)
#define KLARA_PIN(v) ((void)0)
#endif
// Loops over a lane's E elements are fully unrolled (the element arrays are registers).  The run-time compiled closure kernels with
// 256 elements per lane define this to "nounroll" (klara_jit.hip): unrolled, such a kernel takes half a minute to compile per mode
// (tens of thousands of instructions, 512 registers, one wavefront per SIMD) for 12 % more throughput than the loop form over
// scratch-resident arrays, which compiles in a second.
#ifndef KLARA_PRAGMA_UNROLL_E
#define KLARA_PRAGMA_UNROLL_E _Pragma("unroll")
#endif
#ifndef KLARA_E4_WAVES_LOGISTIC
#define KLARA_E4_WAVES_LOGISTIC 2
#endif
#ifndef KLARA_E4_WAVES_LOGISTIC_HMC
#define KLARA_E4_WAVES_LOGISTIC_HMC 2
#endif
#ifndef KLARA_E8_WAVES_LOGISTIC
#define KLARA_E8_WAVES_LOGISTIC 2   /* 5 .. 8 parameters (294 registers when unconstrained): D = 8, 200 rows, 32,768 chains: 1.53e9 against 1.31e9 transitions/s at one wavefront per SIMD, same box */
#endif
#ifndef KLARA_E4_WAVES_PLAIN
#define KLARA_E4_WAVES_PLAIN 3   // the specialised (no tuner, no monitor) E=4 kernels fit 168 VGPRs
#endif
#define KLARA_SLICE_ATT_BITS 14
#define KLARA_SLICE_MAX_ATT ((1 << KLARA_SLICE_ATT_BITS) - 1)
#define KLARA_INIT_TRANSITION ((((uint64_t)1) << 40) - 1)

// Device pointers inside KParams are typed as address space 1 (global) in device compilation: the struct is read
// from memory, and without the qualifier hipcc treats the loaded pointers as generic and emits flat_load/flat_store
// (vmcnt + lgkmcnt, 64-bit VGPR addresses) instead of global_load v, v_off32, s[base:base+1].  Host code sees plain
// pointers of the same size.
#if defined(__HIP_DEVICE_COMPILE__)
#define KGLOBAL __attribute__((address_space(1)))
#else
#define KGLOBAL
#endif
typedef KGLOBAL double gdouble;
typedef KGLOBAL long long glong;
typedef KGLOBAL unsigned long long gulong;
typedef KGLOBAL uint8_t guchar;
typedef KGLOBAL int gint;

// A device-detected error (non-finite start, slice sampler stuck).  The flag is a word of host memory mapped into the device's address
// space (klara_api.hip), so that klara_synchronize reads it without a copy command; a plain store is all a PCIe write offers (no fetch-max), and
// the two conditions cannot meet in one launch (the first is raised by the initialisation kernels, the second by transitions).
__device__ __forceinline__ void klara_raise(gint* flag, int code) { __hip_atomic_store((int*)flag, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// Kernel parameter block.  All pointers are device pointers.
struct KParams {
    gdouble* X; gdouble* GR; gdouble* LT;         // state: value nchains x D, gradlogtarget nchains x D, logtarget nchains
    gdouble* tune_step;                         // per chain (or [0] in pooled mode)
    glong* tune_accepted; glong* tune_proposed; glong* tune_totproposed;
    gulong* pooled_accepted;       // pooled mode: device-wide accepted counter
    guchar* accept;                           // [launch step][nchains] or null
    gulong* naccept;               // per chain
    gdouble* sum; gdouble* sumsq;                // per chain x D or null: running sums in sojourn form (see `held`)
    // Saved (post-burn-in, thinned) steps the chain has spent at its CURRENT state that are not in sum / sumsq yet.  The save
    // rule only counts (held += 1); when a transition is accepted the state being left is folded in first:
    // sum += held * x, sumsq += held * (x * x), held = 0.  Sums over the saved steps = sum + held * x (the read-back kernels
    // form that view).  A chain that did not move during a launch therefore touches neither array.
    glong* held;
    gdouble* hist; long long hist_cols;         // [col][nchains][D] or null
    gdouble* hist_lt; gdouble* hist_g;           // [col][nchains] / [col][nchains][D] or null
    gdouble* hist_ll; gdouble* hist_lp;          // [col][nchains] or null: loglikelihood / logprior of a likelihood + prior user target
    gint* error_flag;                           // set to klara_status on device-detected errors
    long long nchains; long long chain_offset;
    int D; int G; int pooled;
    int rs;                                    // logistic target: data rows split over rs lanes per chain (1 = off)
    unsigned long long seed;
    // sampler
    const gdouble* vecparam;                    // MH sigma[D] / slice widths[D]
    int nleaps; int stepout;
    // tuner
    int tuner; int cnt; double targetrate; double score_k; int period; int is_mh; int tuner_score;
    // DualAveragingMCTuner (KLARA_TUNER_DUAL_AVERAGING): per-chain eps_bar / h_bar arrays and constants
    gdouble* da_epsbar; gdouble* da_hbar; long long da_nadapt; double da_gamma; double da_kappa; int da_t0;
    double da_mu; double da_lambda;            // mu = log(10*leapstep), lambda = nleaps*leapstep (HMC.jl:124-133,192-213)
    double sqrt_step0, inv_step0;              // sqrt(step0), 1/step0 (MALA with an untuned step)
    double step0;                              // initial step (samplers.jl:29-45); the step of every chain when nothing is tuned
    long long burnin; long long thinning; long long nsteps_total;
    // targets
    const gdouble* gw; const gdouble* gmu; double gconst;      // diag (gw/gmu may be null)
    const gdouble* lX; const gdouble* ly; int ndata; double lambda; double lpconst;   // logistic
    const gdouble* hY; const gdouble* hxc; int hR; int hT; double hp0; double ha0; double hb0;   // hierarchical normal
    const gdouble* cdata; long long cndata;                      // user-defined target (klara_custom.h): read-only data block
    // shader-clock probe of the pair-transposed kernels (klara_get_shader_clock): one workgroup in the middle of the grid writes
    // (s_memtime, s_memrealtime) when it starts [2], [3] and when it ends [0], [1] — stores only, nothing kept in registers
    gulong* clock_probe;
};

// Per-launch values, passed by value.  Everything else (KParams) is static for a handle and lives in device memory:
// the kernels read it through a `const KParams* __restrict__` with scalar loads at the point of use, which keeps
// the ~100 dwords of configuration out of the SGPR file (passed by value they were spilled to VGPR lanes:
// 600+ v_readlane/v_writelane in the transition kernel).
struct KLaunch {
    unsigned long long t0;                     // global index of the first transition of this launch
    int nsteps;                                // transitions in this launch
    int save_phase0; long long save_col0;      // save-rule bookkeeping computed on the host (no device division)
    long long group0, group_end;               // layout kind 3: the chain groups [group0, group_end) this launch covers
};

// ------------------------------------------------------------------------------------------------
// cross-lane helpers
// ------------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v)
{
    const uint64_t u = kd_d2u(v);
    // every lane has a valid source under these controls, so no "old" value needs to be materialised
    const int lo = __builtin_amdgcn_mov_dpp((int)(uint32_t)u, CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp((int)(uint32_t)(u >> 32), CTRL, 0xf, 0xf, true);
    return kd_u2d(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}
__device__ __forceinline__ double bperm_xor(double v, int lane, int m)
{
    const uint64_t u = kd_d2u(v);
    const int addr = (lane ^ m) << 2;
    const int lo = __builtin_amdgcn_ds_bpermute(addr, (int)(uint32_t)u);
    const int hi = __builtin_amdgcn_ds_bpermute(addr, (int)(uint32_t)(u >> 32));
    return kd_u2d(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}
// the largest value of v over the wavefront, as a scalar: the trip count of a loop in which every lane has its own (dual averaging: a vote per
// trip — for (l = 0; __any(l < nl); ++l) — costs the register allocator the loop: 100 scratch accesses per leapfrog in k_dense_split, round 6)
__device__ __forceinline__ int wave_max_int(int v)
{
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) { const int o = __builtin_amdgcn_ds_bpermute((lane ^ m) << 2, v); v = o > v ? o : v; }
    return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ double lane_bcast(double v, int src_lane)
{
    const uint64_t u = kd_d2u(v);
    const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(uint32_t)u);
    const int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(uint32_t)(u >> 32));
    return kd_u2d(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}

// xor-butterfly all-reduce over the G lanes of a group; N values at once for ILP.
template <int N>
__device__ __forceinline__ void group_allreduce(double (&v)[N], int G, int lane)
{
    if (G > 1) {
KLARA_PRAGMA_UNROLL_E
        for (int i = 0; i < N; ++i) v[i] = v[i] + dpp_mov<0xB1>(v[i]);    // quad_perm [1,0,3,2]
    }
    if (G > 2) {
KLARA_PRAGMA_UNROLL_E
        for (int i = 0; i < N; ++i) v[i] = v[i] + dpp_mov<0x4E>(v[i]);    // quad_perm [2,3,0,1]
    }
    if (G > 4) {
KLARA_PRAGMA_UNROLL_E
        for (int i = 0; i < N; ++i) v[i] = v[i] + dpp_mov<0x141>(v[i]);   // row_half_mirror
    }
    if (G > 8) {
KLARA_PRAGMA_UNROLL_E
        for (int i = 0; i < N; ++i) v[i] = v[i] + dpp_mov<0x140>(v[i]);   // row_mirror
    }
    if (G > 16) {
KLARA_PRAGMA_UNROLL_E
        for (int i = 0; i < N; ++i) v[i] = v[i] + bperm_xor(v[i], lane, 16);
    }
    if (G > 32) {
KLARA_PRAGMA_UNROLL_E
        for (int i = 0; i < N; ++i) v[i] = v[i] + bperm_xor(v[i], lane, 32);
    }
}

// ------------------------------------------------------------------------------------------------
// lane context
// ------------------------------------------------------------------------------------------------
template <int E>
struct LaneCtx {
    int lane;          // 0..63
    int G;             // lanes per chain
    int q;             // lane within group
    int i0;            // first element index owned (E*q)
    long long chain;   // local chain index (may be >= nchains: inactive group)
    bool chain_ok;
    bool valid[E];     // element i0+e < D and chain_ok
    int RS;            // row split (logistic target): RS lanes share one chain, each holding ALL E elements (G == 1)
    int rq;            // lane within the row-split group
    unsigned voff0;    // byte offset of element i0 inside this wave's chain-group window (group-invariant), see GroupWin
    bool aligned;      // D % E == 0: a lane's E elements are all valid or all padding, and 16-byte aligned
};

template <int E, int GT, bool RSPL = true>
__device__ __forceinline__ LaneCtx<E> make_ctx(const KParams& p)
{
    LaneCtx<E> c;
    const int G = GT ? GT : p.G;
    c.G = G;
    c.lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    c.RS = (RSPL && p.rs > 1) ? p.rs : 1;      // only the logistic target splits data rows over lanes
    c.rq = c.lane & (c.RS - 1);
    c.q = (c.lane / c.RS) & (G - 1);
    const int grp = c.lane / (G * c.RS);
    c.chain = wave * (64 / (G * c.RS)) + grp;
    c.chain_ok = c.chain < p.nchains;
    c.i0 = E * c.q;
KLARA_PRAGMA_UNROLL_E
    for (int e = 0; e < E; ++e) c.valid[e] = c.chain_ok && (c.i0 + e < p.D);
    c.voff0 = (unsigned)((grp * p.D + c.i0) * 8);
    c.aligned = (p.D % E) == 0;
    return c;
}

typedef double kd_double2 __attribute__((ext_vector_type(2)));

template <int E>
__device__ __forceinline__ void load_vec(const LaneCtx<E>& c, const gdouble* base, int D, double (&v)[E])
{
    // out-of-range lanes read element 0 of a valid row (always in bounds) and discard it: a select
    // instead of an exec-masked branch per element
    const gdouble* row = base + (c.chain_ok ? c.chain : 0) * D;
KLARA_PRAGMA_UNROLL_E
    for (int e = 0; e < E; ++e) {
        const double t = row[c.valid[e] ? c.i0 + e : 0];
        v[e] = c.valid[e] ? t : 0.0;
    }
}
template <int E>
__device__ __forceinline__ void store_vec(const LaneCtx<E>& c, gdouble* base, int D, const double (&v)[E])
{
    gdouble* row = base + c.chain * D + c.i0;
KLARA_PRAGMA_UNROLL_E
    for (int e = 0; e < E; ++e) if (c.valid[e] && c.rq == 0) row[e] = v[e];
}
template <int E>
__device__ __forceinline__ void load_param(const LaneCtx<E>& c, const gdouble* base, int D, double dflt,
                                           double (&v)[E])
{
    // (branch-free: E conditional loads are E basic blocks, and at E = 128 the compiler spent two minutes on them)
    if (base == nullptr) {
KLARA_PRAGMA_UNROLL_E
        for (int e = 0; e < E; ++e) v[e] = dflt;
        return;
    }
KLARA_PRAGMA_UNROLL_E
    for (int e = 0; e < E; ++e) {
        const int i = c.i0 + e;
        const double t = base[i < D ? i : D - 1];
        v[e] = i < D ? t : dflt;
    }
}

// ---- chain-group windows ------------------------------------------------------------------------
// The D-vectors of the chains one wavefront works on are contiguous in HBM: [first chain * D, (first + cpw) * D).
// k_transitions addresses them through a buffer resource built per group in SGPRs (base = array + first*D, num_records
// = the bytes of the chains that exist) plus a per-lane byte offset that never changes while the wave walks over its
// groups.  Padding lanes carry an offset past num_records: the hardware returns 0 for their loads and drops their
// stores, and the lanes of chains beyond nchains in the last group fall off the end of the window by themselves — so
// the loop spends no VALU on addresses, bounds selects or exec masks.
#define KLARA_BUF_OOB 0x80000000u
typedef unsigned int kd_uint2 __attribute__((ext_vector_type(2)));
typedef unsigned int kd_uint4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t group_window(const gdouble* base, long long first_chain, int nchains_here, int D)
{
    return __builtin_amdgcn_make_buffer_rsrc((void*)(base + first_chain * D), 0, nchains_here * D * 8, 0x00020000);
}
template <int E>
__device__ __forceinline__ unsigned elem_off(const LaneCtx<E>& c, int D, int e, bool store)
{
    return (c.i0 + e < D && !(store && c.rq != 0)) ? c.voff0 + 8u * (unsigned)e : KLARA_BUF_OOB;
}
template <int E>
__device__ __forceinline__ void load_win(const LaneCtx<E>& c, __amdgpu_buffer_rsrc_t w, int D, double (&v)[E])
{
    if (c.aligned) {
        const unsigned o = c.i0 < D ? c.voff0 : KLARA_BUF_OOB;
KLARA_PRAGMA_UNROLL_E
        for (int e = 0; e < E; e += 2) {
            const kd_uint4 t = __builtin_amdgcn_raw_buffer_load_b128(w, o + 8u * (unsigned)e, 0, 0);
            v[e] = __builtin_bit_cast(double, kd_uint2{ t.x, t.y });
            v[e + 1] = __builtin_bit_cast(double, kd_uint2{ t.z, t.w });
        }
    } else {
KLARA_PRAGMA_UNROLL_E
        for (int e = 0; e < E; ++e)
            v[e] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(w, elem_off<E>(c, D, e, false), 0, 0));
    }
}
template <int E>
__device__ __forceinline__ void store_win(const LaneCtx<E>& c, __amdgpu_buffer_rsrc_t w, int D, const double (&v)[E])
{
    if (c.aligned) {
        const unsigned o = (c.i0 < D && c.rq == 0) ? c.voff0 : KLARA_BUF_OOB;
KLARA_PRAGMA_UNROLL_E
        for (int e = 0; e < E; e += 2) {
            const kd_uint2 a = __builtin_bit_cast(kd_uint2, v[e]), b = __builtin_bit_cast(kd_uint2, v[e + 1]);
            __builtin_amdgcn_raw_buffer_store_b128(kd_uint4{ a.x, a.y, b.x, b.y }, w, o + 8u * (unsigned)e, 0, 0);
        }
    } else {
KLARA_PRAGMA_UNROLL_E
        for (int e = 0; e < E; ++e)
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(kd_uint2, v[e]), w, elem_off<E>(c, D, e, true), 0, 0);
    }
}

// proposal normals of this lane's E elements for transition t: element pair i>>1 <- 64 bits of a block (kd_normal_pair_at),
// cos branch for even i, sin branch for odd i (E is even, i0 is even).
// The accept uniform of a transition is words (x,y) of block slot S = ceil(D/2), the first block past the proposal
// normals.  Whenever the layout has padding (G*E > D, e.g. D = 100 on 32 lanes x 4) some lane evaluates that very
// block as one of its (unused) normal pairs, and Box-Muller has already formed u = kd_u44(x,y) and log(u) for it: the
// accept test then costs one ds_bpermute instead of a Philox block plus a log.  Same words, same kd_log -> same bits.
struct AccDraw { double u, logu; bool have; /* u, logu are this chain's accept draw in every lane already (row-split layout) */ };

// lane k of every quad to the quad's four lanes
__device__ __forceinline__ double quad_bcast(double v, int k)
{
    return k == 0 ? dpp_mov<0x00>(v) : k == 1 ? dpp_mov<0x55>(v) : k == 2 ? dpp_mov<0xAA>(v) : dpp_mov<0xFF>(v);
}

template <int E>
__device__ __forceinline__ void lane_normals(const LaneCtx<E>& c, unsigned long long seed,
                                             unsigned long long gchain, unsigned long long t,
                                             double (&z)[E], AccDraw& ad, int acc_slot, int npairs)
{
    static_assert(E % 2 == 0, "E must be even");
    // Row-split layout (logistic target): RS >= 4 lanes hold the SAME chain and would each evaluate the same E/2 blocks, plus the block
    // of the accept draw.  Lane rq evaluates ONE block instead — slot rq & 3 — and the four lanes of a quad exchange the results: one
    // Philox + Box-Muller evaluation per lane and transition instead of E/2 + 1 (cfg 4, E = 4: 102 + 12 instead of 204 + ~70 vector
    // instructions of a ~2,650-instruction transition).  The same blocks, the same functions: no bit changes.
    if constexpr (E <= 16) {
        if (c.RS >= 4) {
            constexpr int NB = (E / 2 + 3) / 4;        // blocks per lane: 1 up to E = 8, 2 at E = 16 (slots rq & 3 and 4 + (rq & 3))
            double z0[NB], z1[NB], u1[NB], lg1[NB];
#pragma unroll
            for (int m = 0; m < NB; ++m) {      // pair indices < 8: block slot = pair index, words (x, y) (kd_normal_pair_at), real or padding
                const kd_u32x4 b = kd_stream_block(seed, gchain, t, (uint32_t)((c.rq & 3) + 4 * m));
                kd_normal_pair_w(b.x, b.y, &z0[m], &z1[m], &u1[m], &lg1[m]);
            }
KLARA_PRAGMA_UNROLL_E
            for (int j = 0; j < E / 2; ++j) {
                const double a = quad_bcast(z0[j >> 2], j & 3), b = quad_bcast(z1[j >> 2], j & 3);
                z[2 * j] = c.valid[2 * j] ? a : 0.0;
                z[2 * j + 1] = c.valid[2 * j + 1] ? b : 0.0;
            }
            if (acc_slot >= 0 && acc_slot < 4 * NB) {
                const int src = (c.lane & ~3) | (acc_slot & 3);
                double us = u1[0], ls = lg1[0];
#pragma unroll
                for (int m = 1; m < NB; ++m) if ((acc_slot >> 2) == m) { us = u1[m]; ls = lg1[m]; }
                ad.u = lane_bcast(us, src); ad.logu = lane_bcast(ls, src); ad.have = true;
            }
            return;
        }
    }
KLARA_PRAGMA_UNROLL_E
    for (int j = 0; j < E / 2; ++j) {
        double z0, z1, u1, lg1;
        kd_normal_pair_at(seed, gchain, t, (uint32_t)((c.i0 >> 1) + j), (uint32_t)npairs, &z0, &z1, &u1, &lg1);
        if ((c.i0 >> 1) + j == acc_slot) { ad.u = u1; ad.logu = lg1; }
        // padding elements (index >= D, or lanes of a group past the last chain) get z = 0.  With x = g = 0
        // loaded there too, every per-element term downstream (proposal, gradient, kinetic/Metropolis sums)
        // is exactly +-0, so the reductions need no per-term masking: x + 0 keeps the oracle's bits.
        z[2 * j] = c.valid[2 * j] ? z0 : 0.0;
        z[2 * j + 1] = c.valid[2 * j + 1] ? z1 : 0.0;
    }
}

// ------------------------------------------------------------------------------------------------
// targets.  eval<WANT_LT, WANT_GRAD>(ctx, x, ltpart, g): ltpart is the lane partial of the reduction
// term; lt = finalize(group sum of ltpart).
// ------------------------------------------------------------------------------------------------
// KLARA_TARGET_GAUSS_DIAG: lt = c - sum_i w_i (x_i - mu_i)^2, grad_i = (-2 w_i)(x_i - mu_i)
// (README.md:23 `-dot(z,z)`, README.md:155 `-2*z`; test/BasicContMuvParameter.jl:39-56 MvNormal).
template <int E>
struct DiagTarget {
    double w[E], mu[E], c;
    __device__ __forceinline__ void init(const KParams& p, const LaneCtx<E>& cx, double*)
    {
        load_param<E>(cx, p.gw, p.D, 1.0, w);
        load_param<E>(cx, p.gmu, p.D, 0.0, mu);
        c = p.gconst;
    }
    static __device__ __forceinline__ size_t lds_bytes(const KParams&) { return 0; }
    template <bool WANT_LT, bool WANT_GRAD>
    __device__ __forceinline__ void eval(const LaneCtx<E>& cx, const double (&x)[E], double& ltpart,
                                         double (&g)[E]) const
    {
        double acc = 0.0;
KLARA_PRAGMA_UNROLL_E
        for (int e = 0; e < E; ++e) {
            const double dd = x[e] - mu[e];
            if (WANT_LT) acc = acc + w[e] * (dd * dd);   // padding lanes hold x = mu = 0: the term is exactly 0
            if (WANT_GRAD) g[e] = (-2.0 * w[e]) * dd;
        }
        ltpart = acc;
    }
    __device__ __forceinline__ double finalize(double red) const { return c - red; }
};

// KLARA_TARGET_LOGISTIC (doc/examples/swiss/MALA/analytical.jl:11-18), one chain per lane (G == 1):
// the design matrix and outcomes sit in LDS, every lane walks the ndata rows sequentially
// (same-address LDS reads broadcast), so no cross-lane reduction exists.
template <int E>
struct LogisticTarget {
    const double* sX; const double* sy; const double* sL12; int ndata; int D; double lambda, lpconst;
    // The design matrix sits in LDS with a row stride of E doubles, columns D..E-1 zero: a row is read with 16-byte loads at a
    // compile-time stride and the dot product / gradient accumulations run over all E elements without a per-element `e < D` test
    // (fma(0, x, acc) = acc exactly, so the padded terms leave the D-term chains of the oracle untouched).
    static __device__ __forceinline__ size_t lds_bytes(const KParams& p)
    {
        return sizeof(double) * (((size_t)p.ndata * (size_t)(E + 1) + 1) / 2 * 2 + 256);      // rows, responses, (16-byte aligned) kd_log12's table
    }
    __device__ __forceinline__ void init(const KParams& p, const LaneCtx<E>&, double* lds)
    {
        double* X = lds; double* y = lds + (size_t)p.ndata * E;
        for (int i = threadIdx.x; i < p.ndata * E; i += blockDim.x) { const int r = i / E, e = i - r * E; X[i] = e < p.D ? p.lX[r * p.D + e] : 0.0; }
        for (int i = threadIdx.x; i < p.ndata; i += blockDim.x) y[i] = p.ly[i];
        double* l12 = lds + ((size_t)p.ndata * (E + 1) + 1) / 2 * 2;       // the rows' log(1 + t) table: gathered from LDS like the other tables
        for (int i = threadIdx.x; i < 256; i += blockDim.x) l12[i] = kd_l12tab_dev[i];
        __syncthreads();
        sX = X; sy = y; sL12 = l12; ndata = p.ndata; D = p.D; lambda = p.lambda; lpconst = p.lpconst;
    }
    template <bool WANT_LT, bool WANT_GRAD>
    __device__ __forceinline__ void eval(const LaneCtx<E>& cx, const double (&x)[E], double& ltpart,
                                         double (&g)[E]) const
    {
        // rows r = rq, rq + RS, ... of the design matrix belong to this lane (RS = 1: all of them); the RS lane
        // partials are combined by one xor butterfly below (the oracle sums in the same order, layout kind 2)
        double dotxy = 0.0, slog = 0.0, gacc[E];
KLARA_PRAGMA_UNROLL_E
        for (int e = 0; e < E; ++e) gacc[e] = 0.0;
        // One data row: everything a row contributes, in the oracle's order.  The rows of a lane are r = rq, rq + RS, ...: every
        // lane takes ndata / RS of them (a wave-uniform count: a scalar loop the compiler can unroll — the lane-dependent bound
        // `r < ndata` made it a divergent loop with exec-mask bookkeeping and register copies in every iteration) and the lanes
        // with rq < ndata % RS one more.
        // A BATCH of R rows, staged: every row is a chain row -> Xp -> exp (table gather) -> 1 + t -> log (table gather) -> division, i.e.
        // three LDS round trips in a row, and written one row after the other the compiler emits exactly that — a wavefront then issues
        // ~80 vector instructions per row between three exposed waits and relies on its SIMD's other wavefronts alone (VALU busy 0.76 at
        // 4 wavefronts per SIMD, profiles/r3_pmc_kernels.json).  Here the R rows of a batch go through each stage together — all R row reads,
        // all R exp gathers, all R log gathers are in flight before the first is consumed — and the accumulations run last, row by row in
        // ascending order: the operations and the summation order of the one-row form, bit for bit (the oracle's ko_logit_eval).
        const auto rows_of = [&](auto rtag, int r0, auto stride_) {
            const int stride = kstride(stride_);        // (a KInt: compile-time, the batch's rows are reached by immediate offsets)
            constexpr int R = decltype(rtag)::value;
            double row[R][E], yr[R], xp[R], rr[R], th[R], tl[R], t[R], onept[R], invc[R], logc[R], sp[R], lg[R];
            int kk[R]; uint32_t li[R];
#pragma unroll
            for (int j = 0; j < R; ++j) {
                const int r = r0 + j * stride;
KLARA_PRAGMA_UNROLL_E
                for (int e = 0; e < E; ++e) row[j][e] = sX[r * E + e];
                yr[j] = sy[r];
            }
#pragma unroll
            for (int j = 0; j < R; ++j) {
                double a = 0.0;
KLARA_PRAGMA_UNROLL_E
                for (int e = 0; e < E; ++e) a = kd_fma(row[j][e], x[e], a);       // Xp = v[2]*p
                xp[j] = a;
                kd_exp_neg_reduce(__builtin_fabs(a), &kk[j], &rr[j]);             // t = exp(-|Xp|): one exponential for both functions
            }
            KLARA_SCHED_STAGE();
#pragma unroll
            for (int j = 0; j < R; ++j) { const int idx = kk[j] & 127; th[j] = KD_EXPTAB(2 * idx); tl[j] = KD_EXPTAB(2 * idx + 1); }
            KLARA_SCHED_STAGE();
#pragma unroll
            for (int j = 0; j < R; ++j) { KLARA_PIN(rr[j]); rr[j] = kd_exp_neg_poly(rr[j]); }   // (needs no table value: issued under the gathers)
            KLARA_SCHED_STAGE();
#pragma unroll
            for (int j = 0; j < R; ++j) {
                t[j] = kd_exp_neg_combine(kk[j], rr[j], th[j], tl[j]);
                onept[j] = 1.0 + t[j];
                li[j] = kd_log12_bin(onept[j]);
            }
            KLARA_SCHED_STAGE();
#pragma unroll
            for (int j = 0; j < R; ++j) { invc[j] = sL12[2 * li[j]]; logc[j] = sL12[2 * li[j] + 1]; }
            KLARA_SCHED_STAGE();
#pragma unroll
            for (int j = 0; j < R; ++j) lg[j] = kd_div_unit_range(xp[j] >= 0.0 ? 1.0 : t[j], onept[j]);   // 1/(1+exp(-Xp)) (no table value either)
            KLARA_SCHED_STAGE();
#pragma unroll
            for (int j = 0; j < R; ++j) {
                const double l1p = kd_log12_finish(onept[j], invc[j], logc[j]);
                sp[j] = (xp[j] > 0.0 ? xp[j] : 0.0) + l1p;                        // log(1+exp(Xp))
            }
            KLARA_SCHED_STAGE();
#pragma unroll
            for (int j = 0; j < R; ++j) {
                if (WANT_LT) {
                    dotxy = dotxy + xp[j] * yr[j];                                // dot(Xp, v[3])
                    slog = slog + sp[j];                                          // sum(log(1+exp(Xp)))
                }
                if (WANT_GRAD) {
                    const double res = yr[j] - lg[j];                             // v[3]-1./(1+exp(-Xp))
KLARA_PRAGMA_UNROLL_E
                    for (int e = 0; e < E; ++e) gacc[e] = kd_fma(row[j][e], res, gacc[e]);
                }
            }
        };
        // The rows of a lane are r = rq, rq + RS, ...: every lane takes ndata / RS of them (a wave-uniform count: scalar loops) and the
        // lanes with rq < ndata % RS one more.
        constexpr int RB = KLARA_LOGIT_BATCH_OF(E);
        const int nfull = ndata / cx.RS, tail = ndata - nfull * cx.RS;
        const int nbat = nfull / RB;
        if (cx.RS == 4) { for (int b = 0; b < nbat; ++b) rows_of(KInt<RB>(), cx.rq + b * RB * 4, KInt<4>()); }      // (the default split)
        else { for (int b = 0; b < nbat; ++b) rows_of(KInt<RB>(), cx.rq + b * RB * cx.RS, cx.RS); }
        for (int it = nbat * RB; it < nfull; ++it) rows_of(KInt<1>(), cx.rq + it * cx.RS, cx.RS);
        if (cx.rq < tail) rows_of(KInt<1>(), cx.rq + nfull * cx.RS, cx.RS);
        if (cx.RS > 1) {
            double red[E + 2];
            red[0] = dotxy; red[1] = slog;
KLARA_PRAGMA_UNROLL_E
            for (int e = 0; e < E; ++e) red[2 + e] = gacc[e];
            group_allreduce<E + 2>(red, cx.RS, cx.lane);
            dotxy = red[0]; slog = red[1];
KLARA_PRAGMA_UNROLL_E
            for (int e = 0; e < E; ++e) gacc[e] = red[2 + e];
        }
        if (WANT_LT) {
            double dotpp = 0.0;
KLARA_PRAGMA_UNROLL_E
            for (int e = 0; e < E; ++e) dotpp = dotpp + x[e] * x[e];
            const double ll = dotxy - slog;
            const double lp = -0.5 * (dotpp / lambda + lpconst);                  // plogprior
            ltpart = ll + lp;
        }
        if (WANT_GRAD) {
KLARA_PRAGMA_UNROLL_E
            for (int e = 0; e < E; ++e) g[e] = gacc[e] - x[e] / lambda;           // -p/v[1]
        }
    }
    __device__ __forceinline__ double finalize(double red) const { return red; }
};

// KLARA_TARGET_HIER_NORMAL (BUGS "Rats", builder-defined — include/klara_hip.h, oracle ko_hier_eval).
// theta = (a_1, b_1, ..., a_R, b_R, a_c, b_c, s_c, s_a, s_b): a lane's E contiguous elements are E/2 whole rats
// (or part of the 5-element hyper block), so the residual sums of a rat are lane-local and are formed from the rat's
// sufficient statistics (sum y, sum y x, sum y^2 — registers, computed once per launch).  Per evaluation: the five hyper-parameters are broadcast from their
// owner lanes (ds_bpermute), the three precisions exp(-2 s_k) are evaluated ONCE per group (lane l takes k = l % 3,
// results broadcast back) and the five sums over rats go through one 5-value butterfly.
template <int E>
struct HierTarget {
    int R, T, i0, hl[5], he[5];          // owner lane (within the group) and element slot of each hyper-parameter
    double p0, a0, b0;
    // sufficient statistics of the lane's own units and of the centred covariate (oracle ko_hier_eval)
    double Sy[E / 2], Sxy[E / 2], Syy[E / 2], X1, X2, Td;
    static __device__ __forceinline__ size_t lds_bytes(const KParams&) { return 0; }
    __device__ __forceinline__ void init(const KParams& p, const LaneCtx<E>& cx, double*)
    {
        R = p.hR; T = p.hT; p0 = p.hp0; a0 = p.ha0; b0 = p.hb0; i0 = cx.i0;
KLARA_PRAGMA_UNROLL_E
        for (int k = 0; k < 5; ++k) { hl[k] = (2 * R + k) / E; he[k] = (2 * R + k) % E; }
        X1 = 0.0; X2 = 0.0; Td = (double)T;
        for (int j = 0; j < T; ++j) { const double xj = p.hxc[j]; X1 = X1 + xj; X2 = kd_fma(xj, xj, X2); }
KLARA_PRAGMA_UNROLL_E
        for (int pr = 0; pr < E / 2; ++pr) {
            const int ia = i0 + 2 * pr;
            const int rat = ia < 2 * R ? (ia >> 1) : 0;
            double sy = 0.0, sxy = 0.0, syy = 0.0;
            for (int j = 0; j < T; ++j) {
                const double y = p.hY[rat * T + j];
                sy = sy + y; sxy = kd_fma(y, p.hxc[j], sxy); syy = kd_fma(y, y, syy);
            }
            Sy[pr] = sy; Sxy[pr] = sxy; Syy[pr] = syy;
        }
    }
    template <bool WANT_LT, bool WANT_GRAD>
    __device__ __forceinline__ void eval(const LaneCtx<E>& cx, const double (&x)[E], double& ltpart,
                                         double (&g)[E]) const
    {
        const int gb = cx.lane - cx.q;
        // hyper-parameters: a_c, b_c, s_c, s_a, s_b
        double hv[5];
KLARA_PRAGMA_UNROLL_E
        for (int k = 0; k < 5; ++k) {
            double v = 0.0;
KLARA_PRAGMA_UNROLL_E
            for (int e = 0; e < E; ++e) if (e == he[k]) v = x[e];
            hv[k] = (cx.G > 1) ? lane_bcast(v, gb + hl[k]) : v;
        }
        const double ac = hv[0], bc = hv[1], sc = hv[2], sa = hv[3], sb = hv[4];
        double wc, wa, wb;
        if (cx.G >= 4) {
            const int k3 = cx.q % 3;
            const double w = kd_exp(-2.0 * (k3 == 0 ? sc : (k3 == 1 ? sa : sb)));
            wc = lane_bcast(w, gb); wa = lane_bcast(w, gb + 1); wb = lane_bcast(w, gb + 2);
        } else {
            wc = kd_exp(-2.0 * sc); wa = kd_exp(-2.0 * sa); wb = kd_exp(-2.0 * sb);
        }
        double red[5] = { 0.0, 0.0, 0.0, 0.0, 0.0 };     // A1, B1, A2, B2, C2 lane partials
KLARA_PRAGMA_UNROLL_E
        for (int pr = 0; pr < E / 2; ++pr) {
            const int ia = i0 + 2 * pr;                  // element index of a_i; the rat is ia >> 1
            const bool israt = ia < 2 * R;
            const int rat = israt ? (ia >> 1) : 0;
            const double ai = x[2 * pr], bi = x[2 * pr + 1];
            const double da = ai - ac, db = bi - bc;
            // sum r, sum r x, sum r^2 of r_j = y_j - a - b x_j from the unit's sufficient statistics
            const double S1 = kd_fma(-bi, X1, kd_fma(-Td, ai, Sy[pr]));
            const double Sx = kd_fma(-bi, X2, kd_fma(-ai, X1, Sxy[pr]));
            const double u = kd_fma(Td, ai, -2.0 * Sy[pr]);
            const double v = kd_fma(bi, X2, kd_fma(2.0 * ai, X1, -2.0 * Sxy[pr]));
            const double S2 = kd_fma(bi, v, kd_fma(ai, u, Syy[pr]));
            if (WANT_GRAD) { g[2 * pr] = kd_fma(wc, S1, -(wa * da)); g[2 * pr + 1] = kd_fma(wc, Sx, -(wb * db)); }
            // element order within the lane: a-slot terms then b-slot terms, exactly the oracle's term arrays
            red[0] = red[0] + (israt ? da : 0.0);        red[1] = red[1] + 0.0;
            red[2] = red[2] + (israt ? da * da : 0.0);   red[3] = red[3] + 0.0;
            red[4] = red[4] + (israt ? S2 : 0.0);
            red[0] = red[0] + 0.0;                       red[1] = red[1] + (israt ? db : 0.0);
            red[2] = red[2] + 0.0;                       red[3] = red[3] + (israt ? db * db : 0.0);
            red[4] = red[4] + 0.0;
        }
        group_allreduce<5>(red, cx.G, cx.lane);
        const double A1 = red[0], B1 = red[1], A2 = red[2], B2 = red[3], C2 = red[4];
        const double RT = (double)R * (double)T, Rd = (double)R;
        if (WANT_GRAD) {
KLARA_PRAGMA_UNROLL_E
            for (int e = 0; e < E; ++e) {
                const int i = i0 + e;
                if (i == 2 * R) g[e] = kd_fma(wa, A1, -(p0 * ac));
                else if (i == 2 * R + 1) g[e] = kd_fma(wb, B1, -(p0 * bc));
                else if (i == 2 * R + 2) g[e] = kd_fma(2.0 * b0, wc, kd_fma(wc, C2, -RT) - 2.0 * a0);
                else if (i == 2 * R + 3) g[e] = kd_fma(2.0 * b0, wa, kd_fma(wa, A2, -Rd) - 2.0 * a0);
                else if (i == 2 * R + 4) g[e] = kd_fma(2.0 * b0, wb, kd_fma(wb, B2, -Rd) - 2.0 * a0);
                else if (i > 2 * R + 4) g[e] = 0.0;
            }
        }
        if (WANT_LT) {
            const double l_c = (-RT * sc - 0.5 * (wc * C2)) + (-2.0 * a0 * sc - b0 * wc);
            const double l_a = (-Rd * sa - 0.5 * (wa * A2)) + (-2.0 * a0 * sa - b0 * wa);
            const double l_b = (-Rd * sb - 0.5 * (wb * B2)) + (-2.0 * a0 * sb - b0 * wb);
            const double lt = ((l_c + l_a) + l_b) - (0.5 * p0) * (ac * ac + bc * bc);
            ltpart = (cx.q == 0) ? lt : 0.0;            // the caller's butterfly then returns lt itself
        }
    }
    __device__ __forceinline__ double finalize(double red) const { return red; }
};

template <int TARGET, int E> struct TargetSel;
template <int E> struct TargetSel<KLARA_TARGET_GAUSS_DIAG, E> { using type = DiagTarget<E>; };
template <int E> struct TargetSel<KLARA_TARGET_LOGISTIC, E> { using type = LogisticTarget<E>; };
template <int E> struct TargetSel<KLARA_TARGET_HIER_NORMAL, E> { using type = HierTarget<E>; };

__device__ __forceinline__ bool kfinite(double v) { return (v == v) && (v - v == 0.0); }

// full logtarget of a vector (one reduction)
template <class T, int E>
__device__ __forceinline__ double eval_lt(const T& tg, const LaneCtx<E>& cx, const double (&x)[E])
{
    double part[1], gdummy[E];
    tg.template eval<true, false>(cx, x, part[0], gdummy);
    group_allreduce<1>(part, cx.G, cx.lane);
    return tg.finalize(part[0]);
}

// ------------------------------------------------------------------------------------------------
// PLAIN kernels are the instantiation for the common job (VanillaMCTuner, not verbose: nothing counts, nothing tunes):
// the mode flags below fold to constants there, which removes the tuner bookkeeping from the generated code.
#define KCNT (PLAIN ? 0 : p.cnt)
#define KPOOLED (PLAIN ? 0 : p.pooled)
#define KDA (!PLAIN && p.tuner == KLARA_TUNER_DUAL_AVERAGING)

// per-chain tuner state in registers (tuners.jl:5-10), uniform across the group's lanes
// ------------------------------------------------------------------------------------------------
struct TuneRegs { double step; long long accepted, proposed, totproposed; int phase; /* proposed % period */
                  double epsbar, hbar; /* dual averaging */ };

__device__ __forceinline__ void tune_count_proposal(const KParams& p, TuneRegs& tn)
{
    tn.proposed += 1;
    tn.phase = (tn.phase + 1 == p.period) ? 0 : tn.phase + 1;
}

// score functions of AcceptanceRateMCTuner.jl:9,17: logistic(x, 2, k, 0, 0) (stats/logistic.jl:11) or erf(k x) + 1
__device__ __forceinline__ double rate_score(const KParams& p, double x)
{
    if (p.tuner_score == 1) return kd_erf(p.score_k * x) + 1.0;
    return 2.0 / (1.0 + kd_exp(-p.score_k * (x - 0.0))) + 0.0;
}

// tuning block: iterate/MALA.jl:130-152, iterate/HMC.jl:203-224, iterate/MH.jl:116-131;
// rate!/reset_burnin! tuners.jl:27-32; tune! AcceptanceRateMCTuner.jl:46 with logistic_rate_score
// (AcceptanceRateMCTuner.jl:9, stats/logistic.jl:11).
__device__ __forceinline__ void tuning_block(const KParams& p, TuneRegs& tn)
{
    if (!p.cnt) return;
    if (tn.totproposed <= p.burnin && tn.phase == 0) {                // mod(proposed, period) == 0
        const double rate = (double)tn.accepted / (double)tn.proposed;
        if (p.tuner == KLARA_TUNER_ACCEPT_RATE && !p.is_mh) {
            const double xr = rate - p.targetrate;
            tn.step *= rate_score(p, xr);
        }
        tn.totproposed += tn.proposed;
        tn.accepted = 0; tn.proposed = 0; tn.phase = 0;
    }
}

// The same block for kernels that must not branch divergently (klara_dense_big.h): entered when any lane's chain is due, every lane evaluates it, the lanes
// that are not due keep their values through selects — the same operations on the same operands for the lanes that are.
__device__ __forceinline__ void tuning_block_uniform(const KParams& p, TuneRegs& tn)
{
    if (!p.cnt) return;
    const bool due = tn.totproposed <= p.burnin && tn.phase == 0;
    if (!__any(due)) return;
    const double rate = (double)tn.accepted / (double)tn.proposed;
    double step = tn.step;
    if (p.tuner == KLARA_TUNER_ACCEPT_RATE && !p.is_mh) step *= rate_score(p, rate - p.targetrate);
    tn.step = due ? step : tn.step;
    tn.totproposed = due ? tn.totproposed + tn.proposed : tn.totproposed;
    tn.accepted = due ? 0 : tn.accepted; tn.proposed = due ? 0 : tn.proposed;
}

// nleaps of a transition under dual averaging: max(1, Int(round(lambda/step))) — iterate/HMC.jl:142-144
// (round = ties to even; capped at 65536, non-finite quotient -> cap).  Callers pass 1 for the padding lanes of a ragged last wavefront:
// their phantom state (x = 0, re-read every transition) can drive the dual-averaging step towards 0, and the wavefront runs to the
// longest trajectory among its lanes — 65,536 leapfrogs per transition for nothing.
__device__ __forceinline__ int da_nleaps(const KParams& p, double step)
{
    const double q = p.da_lambda / step;
    long long nl = (q == q && q < 65536.0) ? (long long)__builtin_rint(q) : 65536ll;
    return (int)(nl < 1 ? 1 : nl);
}
// tune!(tune, tuner, count, a) — DualAveragingMCTuner.jl:95-101; after nadapt: step = eps_bar (iterate/HMC.jl:247)
__device__ __forceinline__ void da_update(const KParams& p, TuneRegs& tn, long long count, double a)
{
    if (count <= p.da_nadapt) {
        const double hweight = 1.0 / (double)(count + p.da_t0);
        tn.hbar = (1.0 - hweight) * tn.hbar + hweight * (p.targetrate - a);
        tn.step = kd_exp(p.da_mu - __builtin_sqrt((double)count) * tn.hbar / p.da_gamma);
        const double eweight = kd_exp(-p.da_kappa * kd_log((double)count));        // count^(-kappa)
        tn.epsbar = kd_exp((1.0 - eweight) * kd_log(tn.epsbar) + eweight * kd_log(tn.step));
    } else {
        tn.step = tn.epsbar;
    }
}

// ------------------------------------------------------------------------------------------------
// transitions.  Each returns the accept flag (uniform across the group) and updates x, g, lt.
// ------------------------------------------------------------------------------------------------
// accept iff ratio > 0 || ratio > log(rand())  (iterate/MH.jl:97, iterate/MALA.jl:94); `acc` holds ratio > 0.
template <int E>
__device__ __forceinline__ bool accept_log_test(const KParams& p, const LaneCtx<E>& cx, unsigned long long gchain,
                                                unsigned long long t, const AccDraw& ad, bool acc, double ratio)
{
    if (ad.have) return acc || ratio > ad.logu;
    const int acc_owner = ((p.D + 1) >> 1) / (E / 2);          // lane (within the group) whose normals used slot ceil(D/2)
    if (acc_owner < cx.G) {                                    // free: log(u) was formed by that lane's Box-Muller
        const double logu = cx.G > 1 ? lane_bcast(ad.logu, (cx.lane - cx.q) + acc_owner) : ad.logu;
        return acc || ratio > logu;
    }
    if (!acc && ratio > KD_LOG_UMIN_GUARD) {   // below the guard no uniform of the stream can accept
        const double u = kd_accept_uniform(kd_stream_block(p.seed, gchain, t, (uint32_t)((p.D + 1) >> 1)));
        acc = ratio > kd_log_u01(u);
    }
    return acc;
}

// What a transition proposed.  With COMMIT = false (one transition per launch, nothing monitored) the step functions
// leave the chain registers alone and hand the proposal back, and the kernel writes an accepted proposal straight from
// these registers to HBM — no conditional register copies.
template <int E>
struct Proposal { double x[E], g[E], lt; };

// iterate!(job, MH, Multivariate) — iterate/MH.jl:72-124 (symmetric normalised branch)
template <class T, int E, bool COMMIT = true>
__device__ __forceinline__ bool step_mh(const KParams& p, const T& tg, const LaneCtx<E>& cx,
                                        unsigned long long gchain, unsigned long long t,
                                        const double (&z)[E], const AccDraw& ad, const double (&sigma)[E],
                                        double (&x)[E], double& lt, Proposal<E>& prop)
{
    double xp[E], gd[E], red[1];
KLARA_PRAGMA_UNROLL_E
    for (int e = 0; e < E; ++e) xp[e] = x[e] + sigma[e] * z[e];                       // MH.jl:79
    tg.template eval<true, false>(cx, xp, red[0], gd);                                // :81
    group_allreduce<1>(red, cx.G, cx.lane);
    const double ltp = tg.finalize(red[0]);
    const double ratio = ltp - lt;                                                    // :83
    bool acc = ratio > 0.0;                                                           // :97
    acc = accept_log_test<E>(p, cx, gchain, t, ad, acc, ratio);
    if (!COMMIT) {
KLARA_PRAGMA_UNROLL_E
        for (int e = 0; e < E; ++e) prop.x[e] = xp[e];
        prop.lt = ltp;
    } else if (acc) {                                                                 // :98-100
KLARA_PRAGMA_UNROLL_E
        for (int e = 0; e < E; ++e) x[e] = xp[e];
        lt = ltp;
    }
    return acc;
}

// iterate!(job, MALA, Multivariate) — iterate/MALA.jl:78-128
template <class T, int E, bool PLAIN, bool COMMIT = true>
__device__ __forceinline__ bool step_mala(const KParams& p, const T& tg, const LaneCtx<E>& cx,
                                          unsigned long long gchain, unsigned long long t,
                                          const double (&z)[E], const AccDraw& ad, double h,
                                          double (&x)[E], double (&g)[E], double& lt, Proposal<E>& prop)
{
    double mu[E], xp[E], gp[E], red[3];
    // sqrt(step) and 1/step come precomputed from the host while nothing tunes the step (same IEEE results)
    const double halfh = 0.5 * h, sq = KCNT ? __builtin_sqrt(h) : p.sqrt_step0;
    // abs2(.)/step of MALA.jl:90,92 is evaluated as abs2(.) * (1/step): one f64 division per transition instead
    // of 2 per element (a division is ~70 issue cycles per wave on gfx950); the oracle does the same.
    const double inv_h = KCNT ? 1.0 / h : p.inv_step0;
    const double half_inv_h = 0.5 * inv_h;                           // 0.5*(abs2(.)*inv_h) == abs2(.)*(0.5*inv_h): halving is exact
KLARA_PRAGMA_UNROLL_E
    for (int e = 0; e < E; ++e) mu[e] = x[e] + halfh * g[e];                          // :83
KLARA_PRAGMA_UNROLL_E
    for (int e = 0; e < E; ++e) xp[e] = mu[e] + sq * z[e];                            // :84
    // (:90 does not depend on the evaluation: formed first, so that the drift mean is not live across it)
    double s1 = 0.0, s2 = 0.0;
KLARA_PRAGMA_UNROLL_E
    for (int e = 0; e < E; ++e) {
        const double q1 = mu[e] - xp[e];
        s1 = s1 + (q1 * q1) * half_inv_h;                        // :90
    }
    tg.template eval<true, true>(cx, xp, red[0], gp);                                 // :86
KLARA_PRAGMA_UNROLL_E
    for (int e = 0; e < E; ++e) {
        const double mup = xp[e] + halfh * gp[e];                                     // :91
        const double q2 = mup - x[e];
        s2 = s2 + (q2 * q2) * half_inv_h;                        // :92
    }
    red[1] = s1; red[2] = s2;
    group_allreduce<3>(red, cx.G, cx.lane);
    const double ltp = tg.finalize(red[0]);
    double ratio = ltp - lt;                                                          // :88
    ratio += red[1];                                                                  // :90
    ratio -= red[2];                                                                  // :92
    bool acc = ratio > 0.0;                                                           // :94
    acc = accept_log_test<E>(p, cx, gchain, t, ad, acc, ratio);
    if (!COMMIT) {
KLARA_PRAGMA_UNROLL_E
        for (int e = 0; e < E; ++e) { prop.x[e] = xp[e]; prop.g[e] = gp[e]; }
        prop.lt = ltp;
    } else if (acc) {                                                                 // :95-105
KLARA_PRAGMA_UNROLL_E
        for (int e = 0; e < E; ++e) { x[e] = xp[e]; g[e] = gp[e]; }
        lt = ltp;
    }
    return acc;
}

// iterate!(job, HMC, Multivariate) — iterate/HMC.jl:124-201; leapfrog! samplers.jl:122-134;
// hamiltonian samplers.jl:103
template <class T, int E, bool PLAIN, bool COMMIT = true>
__device__ __forceinline__ bool step_hmc(const KParams& p, const T& tg, const LaneCtx<E>& cx,
                                         unsigned long long gchain, unsigned long long t,
                                         const double (&z)[E], const AccDraw& ad, double eps, int nleaps, double& a_out,
                                         double (&x)[E], double (&g)[E], double& lt, Proposal<E>& prop)
{
    double mom[E], xp[E], gp[E], red[2], dummy;
KLARA_PRAGMA_UNROLL_E
    for (int e = 0; e < E; ++e) mom[e] = z[e];                                        // :135
    double k0[1] = { 0.0 };
KLARA_PRAGMA_UNROLL_E
    for (int e = 0; e < E; ++e) k0[0] = k0[0] + mom[e] * mom[e];
    group_allreduce<1>(k0, cx.G, cx.lane);
    const double H0 = lt - 0.5 * k0[0];                                               // :137
KLARA_PRAGMA_UNROLL_E
    for (int e = 0; e < E; ++e) { xp[e] = x[e]; gp[e] = g[e]; }                       // :139-140
    const double halfe = 0.5 * eps;
    // leapfrog! `nleaps` times (:146-155, samplers.jl:122-134) in its merged form (deliberate deviation (7); the
    // oracle takes the same steps): the closing half-kick of step l and the opening half-kick of step l + 1 use the same gradient and
    // are ONE update p += eps g, and every update is one fma:
    //   p = fma(eps/2, g, p);  L x { x = fma(eps, p, x);  g = grad(x);  p = fma(l < L-1 ? eps : eps/2, g, p) }
KLARA_PRAGMA_UNROLL_E
    for (int e = 0; e < E; ++e) mom[e] = kd_fma(halfe, gp[e], mom[e]);
    if (!KDA) {
        const int nl = p.nleaps;
        for (int l = 0; l < nl; ++l) {
KLARA_PRAGMA_UNROLL_E
            for (int e = 0; e < E; ++e) xp[e] = kd_fma(eps, mom[e], xp[e]);
            tg.template eval<false, true>(cx, xp, dummy, gp);
            const double kf = l + 1 < nl ? eps : halfe;
KLARA_PRAGMA_UNROLL_E
            for (int e = 0; e < E; ++e) mom[e] = kd_fma(kf, gp[e], mom[e]);
        }
    } else {
        // dual averaging: the trip count differs per chain (iterate/HMC.jl:142-144); the wavefront runs to the
        // longest trajectory it carries and finished chains keep their state (the target evaluation may use
        // cross-lane collectives, so control flow stays wave-uniform)
        const int nlmax = wave_max_int(nleaps);
        for (int l = 0; l < nlmax; ++l) {
            const bool go = l < nleaps;
            double gn[E];
KLARA_PRAGMA_UNROLL_E
            for (int e = 0; e < E; ++e) { const double xn = kd_fma(eps, mom[e], xp[e]); xp[e] = go ? xn : xp[e]; }
            tg.template eval<false, true>(cx, xp, dummy, gn);
            const double kf = l + 1 < nleaps ? eps : halfe;          // the chain's own last step closes with a half-kick
KLARA_PRAGMA_UNROLL_E
            for (int e = 0; e < E; ++e) {
                gp[e] = go ? gn[e] : gp[e];
                const double mn = kd_fma(kf, gp[e], mom[e]);
                mom[e] = go ? mn : mom[e];
            }
        }
    }
    double gd[E];
    tg.template eval<true, false>(cx, xp, red[0], gd);                                // :157
    double k1 = 0.0;
KLARA_PRAGMA_UNROLL_E
    for (int e = 0; e < E; ++e) k1 = k1 + mom[e] * mom[e];
    red[1] = k1;
    group_allreduce<2>(red, cx.G, cx.lane);
    const double ltp = tg.finalize(red[0]);
    const double H1 = ltp - 0.5 * red[1];                                             // :159
    const double ratio = H1 - H0;                                                     // :161
    const double ex = kd_exp(ratio);
    const double a = 1.0 < ex ? 1.0 : ex;                                             // :163
    a_out = a;
    const int acc_owner = ((p.D + 1) >> 1) / (E / 2);
    const double u = ad.have ? ad.u : (acc_owner < cx.G)
        ? (cx.G > 1 ? lane_bcast(ad.u, (cx.lane - cx.q) + acc_owner) : ad.u)
        : kd_accept_uniform(kd_stream_block(p.seed, gchain, t, (uint32_t)((p.D + 1) >> 1)));
    const bool acc = u < a;                                                           // :165
    if (!COMMIT) {
KLARA_PRAGMA_UNROLL_E
        for (int e = 0; e < E; ++e) { prop.x[e] = xp[e]; prop.g[e] = gp[e]; }
        prop.lt = ltp;
    } else if (acc) {                                                                 // :166-176
KLARA_PRAGMA_UNROLL_E
        for (int e = 0; e < E; ++e) { x[e] = xp[e]; g[e] = gp[e]; }
        lt = ltp;
    }
    return acc;
}

// iterate!(job, SliceSampler, Multivariate) — iterate/SliceSampler.jl:60-109 — with THE CHAINS OF A WAVEFRONT OUT OF LOCKSTEP (round 5).
// A probe is a full evaluation of the log-target (the reference calls logtarget! on the whole vector, :77-94), made by all chains of the wavefront
// at once; a chain's result depends on its own vector only, and every draw is addressed by (transition, coordinate, attempt).  So nothing obliges
// the chains to probe the same coordinate or the same stage of its update: each chain is a little machine (start of a coordinate -> step-out to the
// left -> to the right -> shrink attempts -> next coordinate) that takes ONE probe per pass, whichever its stage asks for, and the transition ends
// when the slowest chain of the wavefront has updated its D coordinates (probe counts add up over the coordinates, so the spread is a few per cent;
// rounds 1-4 ran every loop until the slowest chain of the wavefront was through it, at five call sites of the evaluation).  The candidate is
// written into the owner lane's register in place; an accepted candidate simply stays there.
//   read_coord(i, x_i, w_i): value and width of coordinate i of the lane's chain (the same in all lanes of the chain);
//   place(i, on, cand): the owner lane of coordinate i writes cand into its register when `on`;   probe(): log-target of the lane's chain.
// `cur` is the chain's log-target (in: current, out: new); `stuck` is sticky (KLARA_ERR_SLICE_STUCK: the chain stays at the state before the update that failed).
template <class ReadCoord, class Place, class Probe>
__device__ __forceinline__ void slice_free_machine(const KParams& p, bool chain_ok, unsigned long long gchain, unsigned long long t,
                                                   double& cur, bool& stuck, ReadCoord read_coord, Place place, Probe probe)
{
    int i = 0, ph = 0;                                   // coordinate; stage: 0 start, 1 step-out left, 2 step-out right, 3 shrink
    bool active = chain_ok && !stuck && p.D > 0;
    double Li = 0.0, Ri = 0.0, logu = 0.0, xi = 0.0, wd = 0.0;
    uint32_t a = 1, guard = 0;
    bool stuck_here = false;                             // ... in this transition (a chain that was stuck before does not enter the loop)
    while (__any(active)) {
        const int ic = i < p.D ? i : p.D - 1;
        const uint32_t base = (uint32_t)ic << KLARA_SLICE_ATT_BITS;
        {   // a chain at the start of coordinate i (:65-73; executed by all, kept by the starting ones)
            const bool starting = active && ph == 0;
            double xs, ws;
            read_coord(ic, xs, ws);
            const kd_u32x4 b0 = kd_stream_block(p.seed, gchain, t, base);
            const double lus = kd_log_u01(kd_uniform_xy(b0)) + cur;                            // :66
            const double ru = kd_uniform_zw(b0);                                               // :71
            xi = starting ? xs : xi; wd = starting ? ws : wd; logu = starting ? lus : logu;
            Li = starting ? xs - ru * ws : Li;                                                 // :72
            Ri = starting ? xs + (1.0 - ru) * ws : Ri;                                         // :73
            a = starting ? 1u : a; guard = starting ? 0u : guard;
            ph = starting ? (p.stepout ? 1 : 3) : ph;
        }
        const double u = kd_slice_attempt_uniform(p.seed, gchain, t, base, a);
        const double cand = ph == 1 ? Li : (ph == 2 ? Ri : u * (Ri - Li) + Li);                // :76 / :83 / :92-93
        place(ic, active, cand);
        const double lc = probe();                                                             // :77 / :84 / :94
        const bool above = lc > logu;
        // step-out (:75-89): while the end is inside the slice, move it out by one width and probe again
        const bool out = active && ph != 3 && above;
        guard += out ? 1u : 0u;
        const bool over = out && guard > (uint32_t)KLARA_SLICE_MAX_ATT;
        Li = (out && !over && ph == 1) ? Li - wd : Li;
        Ri = (out && !over && ph == 2) ? Ri + wd : Ri;
        const bool next_stage = active && ph != 3 && !above;
        // shrink (:91-106)
        const bool shr = active && ph == 3;
        const bool acc = shr && above;                                                         // :95
        const bool rej = shr && !above;
        Ri = (rej && cand > xi) ? cand : Ri;                                                   // :98
        Li = (rej && cand < xi) ? cand : Li;                                                   // :100
        const bool nowhere = rej && !(cand > xi) && !(cand < xi);                              // :102
        a += rej ? 1u : 0u;
        const bool spent = rej && a > (uint32_t)KLARA_SLICE_MAX_ATT;
        stuck = stuck || over || nowhere || spent;
        stuck_here = stuck_here || over || nowhere || spent;
        cur = acc ? lc : cur;                                                                  // :108 (the candidate is already in its register)
        guard = next_stage ? 0u : guard;
        ph = next_stage ? ph + 1 : (acc ? 0 : ph);
        i += acc ? 1 : 0;
        active = active && !stuck && i < p.D;
    }
    // A chain that ran out of attempts holds a rejected candidate (or a step-out end point) in the coordinate it was updating while `cur` is still the
    // log-target of the state before that update: the coordinate goes back to its value, so that (X, LT) read after KLARA_ERR_SLICE_STUCK is a state
    // the chain was in — the one before the update that failed (outside the pass loop: nothing per probe)
    if (__any(stuck_here)) place(i < p.D ? i : p.D - 1, stuck_here, xi);
}

// the group layout: coordinate i lives on lane qo = i / E of its chain's G lanes as register eo = i - qo E
template <class T, int E>
__device__ __forceinline__ bool step_slice_free(const KParams& p, const T& tg, const LaneCtx<E>& cx,
                                           unsigned long long gchain, unsigned long long t,
                                           const double (&widths)[E], double (&x)[E], double& lt,
                                           bool& stuck)
{
    const int group_base = cx.lane - cx.q;
    slice_free_machine(p, cx.chain_ok, gchain, t, lt, stuck,
        [&](int i, double& xs, double& ws) {
            const int qo = i / E, eo = i - qo * E;
            double xi_l = 0.0, w_l = 0.0;
KLARA_PRAGMA_UNROLL_E
            for (int e = 0; e < E; ++e) { xi_l = (e == eo) ? x[e] : xi_l; w_l = (e == eo) ? widths[e] : w_l; }
            xs = (cx.G > 1) ? lane_bcast(xi_l, group_base + qo) : xi_l;
            ws = (cx.G > 1) ? lane_bcast(w_l, group_base + qo) : w_l;
        },
        [&](int i, bool on, double cand) {
            const int qo = i / E, eo = i - qo * E;
            const bool owner = on && cx.q == qo;
KLARA_PRAGMA_UNROLL_E
            for (int e = 0; e < E; ++e) x[e] = (owner && e == eo) ? cand : x[e];
        },
        [&]() { return eval_lt<T, E>(tg, cx, x); });
    return true;
}

// The loops in lockstep (rounds 1-4): the step-out and shrink loops run until every chain of the wavefront is through them (wave-uniform control
// flow via __any; finished chains are masked).  Kept for the jobs where it measures faster, see step_slice below.
template <class T, int E>
__device__ __forceinline__ bool step_slice_lockstep(const KParams& p, const T& tg, const LaneCtx<E>& cx,
                                           unsigned long long gchain, unsigned long long t,
                                           const double (&widths)[E], double (&x)[E], double& lt,
                                           bool& stuck)
{
    const int group_base = cx.lane - cx.q;
    for (int i = 0; i < p.D; ++i) {                                                   // :65
        const int qo = i / E, eo = i - qo * E;
        const bool owner = (cx.q == qo);
        // coordinate value and width, broadcast from the owner lane
        double xi_l = 0.0, w_l = 0.0;
KLARA_PRAGMA_UNROLL_E
        for (int e = 0; e < E; ++e) if (e == eo) { xi_l = x[e]; w_l = widths[e]; }
        const double xi = (cx.G > 1) ? lane_bcast(xi_l, group_base + qo) : xi_l;
        const double w = (cx.G > 1) ? lane_bcast(w_l, group_base + qo) : w_l;
        const uint32_t base = (uint32_t)i << KLARA_SLICE_ATT_BITS;
        const kd_u32x4 b0 = kd_stream_block(p.seed, gchain, t, base);
        const double logu = kd_log_u01(kd_uniform_xy(b0)) + lt;                           // :66
        const double ru = kd_uniform_zw(b0);                                          // :71
        double Li = xi - ru * w;                                                      // :72
        double Ri = xi + (1.0 - ru) * w;                                              // :73
        double tmp[E];
        auto lt_with = [&](double cand) -> double {
KLARA_PRAGMA_UNROLL_E
            for (int e = 0; e < E; ++e) tmp[e] = (owner && e == eo) ? cand : x[e];
            return eval_lt<T, E>(tg, cx, tmp);
        };
        if (p.stepout) {                                                              // :75-89
            double l = lt_with(Li);
            int guard = 0;
            while (true) {
                bool go = cx.chain_ok && !stuck && (l > logu);
                if (go && ++guard > KLARA_SLICE_MAX_ATT) { stuck = true; go = false; }
                if (!__any(go)) break;
                const double Ln = Li - w;
                const double ln = lt_with(go ? Ln : Li);
                if (go) { Li = Ln; l = ln; }
            }
            double r = lt_with(Ri);
            guard = 0;
            while (true) {
                bool go = cx.chain_ok && !stuck && (r > logu);
                if (go && ++guard > KLARA_SLICE_MAX_ATT) { stuck = true; go = false; }
                if (!__any(go)) break;
                const double Rn = Ri + w;
                const double rn = lt_with(go ? Rn : Ri);
                if (go) { Ri = Rn; r = rn; }
            }
        }
        double xprime = xi, ltnew = lt;
        bool done = !cx.chain_ok || stuck;
        for (uint32_t a = 1;; ++a) {                                                  // :91-106
            if (!done && a > KLARA_SLICE_MAX_ATT) { stuck = true; done = true; }
            if (!__any(!done)) break;
            const double u = kd_slice_attempt_uniform(p.seed, gchain, t, base, a);
            const double cand = u * (Ri - Li) + Li;                                   // :92-93
            const double lc = lt_with(done ? xprime : cand);                          // :94
            if (!done) {
                xprime = cand; ltnew = lc;
                if (lc > logu) done = true;                                           // :95
                else if (cand > xi) Ri = cand;                                        // :98
                else if (cand < xi) Li = cand;                                        // :100
                else { stuck = true; done = true; }                                   // :102
            }
        }
        if (!stuck) {
            lt = ltnew;
KLARA_PRAGMA_UNROLL_E
            for (int e = 0; e < E; ++e) if (owner && e == eo) x[e] = xprime;          // :108
        }
    }
    return true;
}

// Which form runs (same box, profiles/r5_ab_slice_group.txt): the free-running machine spends ~150 instructions per pass on its own bookkeeping (the
// starting block and the attempt's block are formed in every pass, because some chain of the wavefront needs them in every pass), the lockstep form
// wastes passes (the slowest of the wavefront's chains at every stage).  Without step-out (one stage: the waste is all in the shrink loop) the machine
// wins on every target tried (+1 % .. +35 %); with step-out it wins where a probe is expensive next to the bookkeeping — the logistic regression's
// pass over its data rows: +37 % — and loses on cheap probes (-14 % .. -28 %), which therefore keep the lockstep form.
template <class T> struct SliceProbeCostly { static constexpr bool value = false; };
template <int E> struct SliceProbeCostly<LogisticTarget<E>> { static constexpr bool value = true; };
template <class T, int E>
__device__ __forceinline__ bool step_slice(const KParams& p, const T& tg, const LaneCtx<E>& cx,
                                           unsigned long long gchain, unsigned long long t,
                                           const double (&widths)[E], double (&x)[E], double& lt,
                                           bool& stuck)
{
    if (SliceProbeCostly<T>::value || !p.stepout) return step_slice_free<T, E>(p, tg, cx, gchain, t, widths, x, lt, stuck);
    return step_slice_lockstep<T, E>(p, tg, cx, gchain, t, widths, x, lt, stuck);
}

// ------------------------------------------------------------------------------------------------
// the transition kernel: run(job) loop of BasicMCJob.jl:219-238 for p.nsteps transitions
// ------------------------------------------------------------------------------------------------
// Per-chain registers that travel between HBM and the transition loop.
template <int E>
struct ChainRegs {
    double x[E], g[E];
    double lt;
    double step; long long accepted, proposed, totproposed;
    double epsbar, hbar;
};

template <int E, bool NEEDG, bool PLAIN>
__device__ __forceinline__ void load_chain(const KParams& p, const LaneCtx<E>& cx, ChainRegs<E>& r, long long first_chain, int here)
{
    load_win<E>(cx, group_window(p.X, first_chain, here, p.D), p.D, r.x);
    if (NEEDG) load_win<E>(cx, group_window(p.GR, first_chain, here, p.D), p.D, r.g);
    const long long c0 = cx.chain_ok ? cx.chain : 0;
    r.lt = p.LT[c0];
    if (KCNT && !KPOOLED) {              // per-chain tuner state (tuners.jl:5-10) only when something counts
        r.step = p.tune_step[c0]; r.accepted = p.tune_accepted[c0];
        r.proposed = p.tune_proposed[c0]; r.totproposed = p.tune_totproposed[c0];
    } else if (KDA) {
        r.step = p.tune_step[c0];
    }
    if (KDA) { r.epsbar = p.da_epsbar[c0]; r.hbar = p.da_hbar[c0]; }
}

template <int E, int GT, bool RSPL = true>
__device__ __forceinline__ void set_chain(const KParams& p, LaneCtx<E>& c, long long group_index)
{
    const int W = RSPL ? c.G * c.RS : c.G;
    c.chain = group_index * (64 / W) + (c.lane / W);
    c.chain_ok = c.chain < p.nchains;
KLARA_PRAGMA_UNROLL_E
    for (int e = 0; e < E; ++e) c.valid[e] = c.chain_ok && (c.i0 + e < p.D);
}

// The transition kernel: run(job) loop of BasicMCJob.jl:219-238 for p.nsteps transitions of every chain.
//
// A wavefront is persistent over several chain groups (group index = wave + k * total_waves): the HBM
// loads of the NEXT group are issued before the current group's transitions are computed, and the
// first step's proposal normals (Philox + Box-Muller, independent of the state) are generated before
// the loaded registers are first touched, so HBM latency overlaps the RNG/ALU work instead of
// serialising with it (one-launch-per-transition mode is otherwise latency-bound at 3 waves/SIMD).
// MODE bit 0 (PLAIN): nothing counts / tunes.  MODE bit 1 (NOMON): no monitor at all (no accept mask, running sums or
// history) — the save-rule bookkeeping disappears from the generated code.
template <int SAMPLER, int TARGET, int E, int GT, int MODE>
// (MALA and HMC on the logistic target at E = 4 — cfg 4 — ask for 4 wavefronts per SIMD: its row loop is a chain of exp / log / division latencies that two
//  wavefronts cannot cover; the 128-register budget spills 156-272 B outside the row loop and still measured 1.01e9 against 8.1e8
//  transitions/s with running sums, 1.05e9 against 9.4e8 without, same box)
__global__ __launch_bounds__(256, (TARGET == KLARA_TARGET_CUSTOM && GT > 1 ? 2 /* staged closures: two workgroups' rows fit a CU's LDS */ :
                                   E == 2 ? (TARGET == KLARA_TARGET_GAUSS_DIAG ? KLARA_E2_DIAG_WAVES : 3) : (E == 4 ? (TARGET == KLARA_TARGET_LOGISTIC && SAMPLER == KLARA_SAMPLER_MALA ? KLARA_E4_WAVES_LOGISTIC
                                                                  : TARGET == KLARA_TARGET_LOGISTIC && SAMPLER == KLARA_SAMPLER_HMC ? KLARA_E4_WAVES_LOGISTIC_HMC
                                                                  : ((MODE & 3) == 3 ? KLARA_E4_WAVES_PLAIN : KLARA_E4_WAVES))
                                                  : (E == 8 && TARGET == KLARA_TARGET_LOGISTIC ? KLARA_E8_WAVES_LOGISTIC : 1))))
void k_transitions(const KParams* __restrict__ pp, const KLaunch kl)
{
    constexpr bool PLAIN = (MODE & 1) != 0, NOMON = (MODE & 2) != 0;
    constexpr bool ONESTEP = (MODE & 4) != 0;          // exactly one transition per launch (= one iterate!)
    const int nsteps = ONESTEP ? 1 : kl.nsteps;
    constexpr bool RSPL = TARGET == KLARA_TARGET_LOGISTIC;
    // single unmonitored transition: an accepted proposal goes from the proposal registers straight to HBM
    constexpr bool DIRECT = ONESTEP && NOMON && SAMPLER != KLARA_SAMPLER_SLICE;
    // monitored jobs: the step functions hand the proposal back and the commit happens here, after the state being left has been
    // folded into the running sums (KParams::held)
    constexpr bool OUTER = !NOMON && SAMPLER != KLARA_SAMPLER_SLICE;
    // The logistic targets fold the running sums of a chain that moves straight into memory (a read-modify-write of 2 D values per
    // accepted transition, against ~3,000 vector instructions of a transition) instead of keeping 4 E registers of sums resident:
    // with them the 128-register kernels of cfg 4 (4 wavefronts per SIMD) spilled.  Everything else keeps resident sums (the README
    // job runs a transition in ~60 instructions: a memory round trip per accepted move would dominate it).
    constexpr bool MEMSUMS = TARGET == KLARA_TARGET_LOGISTIC;
    const KParams& p = *pp;
    kd_tables_to_lds();
    guchar* const accept_out = (!NOMON && p.accept != nullptr) ? p.accept + kl.t0 * (unsigned long long)p.nchains : nullptr;
    gdouble* const hist = NOMON ? nullptr : p.hist;
    gdouble* const hist_lt = NOMON ? nullptr : p.hist_lt;
    gdouble* const hist_g = NOMON ? nullptr : p.hist_g;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using T = typename TargetSel<TARGET, E>::type;
    constexpr bool NEEDG = (SAMPLER == KLARA_SAMPLER_MALA || SAMPLER == KLARA_SAMPLER_HMC);
    constexpr bool NEEDZ = (SAMPLER != KLARA_SAMPLER_SLICE);

    LaneCtx<E> cx = make_ctx<E, GT, RSPL>(p);
    T tg;
    tg.init(p, cx, reinterpret_cast<double*>(smem));
    double vp[E];
    if (SAMPLER == KLARA_SAMPLER_MH || SAMPLER == KLARA_SAMPLER_SLICE) load_param<E>(cx, p.vecparam, p.D, 1.0, vp);

    const int cpw = 64 / (cx.G * cx.RS);
    const long long nwaves = (long long)gridDim.x * (blockDim.x >> 6);
    long long grp = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    const bool do_sum = !NOMON && p.sum != nullptr;
    const bool per_chain_tune = KCNT && !KPOOLED;
    const bool da = KDA;

    // chains of group g: [g*cpw, g*cpw + here), here = the ones that exist (the last group may be short)
    const auto here_of = [&](long long g) { const long long left = p.nchains - g * cpw; return left < cpw ? (left > 0 ? (int)left : 0) : cpw; };
    ChainRegs<E> cur;
    set_chain<E, GT, RSPL>(p, cx, grp);
    load_chain<E, NEEDG, PLAIN>(p, cx, cur, grp * cpw, here_of(grp));

    while (true) {
        // One transition per launch: the next group this wave owns is prefetched (its load latency would otherwise be a visible part
        // of the launch).  Fused launches amortise that latency over their transitions and load the next group when they get to it —
        // the prefetched copy would hold 2 E + 8 registers through every transition (logistic MALA at 4 wavefronts per SIMD: the
        // difference between fitting 128 registers and spilling).
        constexpr bool PREFETCH = ONESTEP;
        const long long grp_next = grp + nwaves;
        const bool has_next = grp_next * cpw < p.nchains;
        LaneCtx<E> cxn = cx;
        ChainRegs<E> nxt;
        if (PREFETCH && has_next) {
            set_chain<E, GT, RSPL>(p, cxn, grp_next);
            load_chain<E, NEEDG, PLAIN>(p, cxn, nxt, grp_next * cpw, here_of(grp_next));
        }
        const long long first_chain = grp * cpw;
        const int here = here_of(grp);

        const unsigned long long gchain = (unsigned long long)(p.chain_offset + cx.chain);
        // running sums are first needed at the end of a transition: not prefetched (saves 4E VGPRs)
        double sm[MEMSUMS ? 1 : E], sq[MEMSUMS ? 1 : E];
        long long held = 0;
        if (do_sum) {
            if constexpr (!MEMSUMS) {
                load_win<E>(cx, group_window(p.sum, first_chain, here, p.D), p.D, sm);
                load_win<E>(cx, group_window(p.sumsq, first_chain, here, p.D), p.D, sq);
            }
            held = p.held[cx.chain_ok ? cx.chain : 0];
        }
        // the fold of the state being left (KParams::held) — into the resident sums, or (MEMSUMS) straight into memory
        const auto fold_state = [&]() {
            const double hf = (double)held;
            if constexpr (MEMSUMS) {
                double ts[E], tq[E];
                const __amdgpu_buffer_rsrc_t ws = group_window(p.sum, first_chain, here, p.D), wq = group_window(p.sumsq, first_chain, here, p.D);
                load_win<E>(cx, ws, p.D, ts); load_win<E>(cx, wq, p.D, tq);
KLARA_PRAGMA_UNROLL_E
                for (int e = 0; e < E; ++e) { ts[e] = ts[e] + hf * cur.x[e]; tq[e] = tq[e] + hf * (cur.x[e] * cur.x[e]); }
                store_win<E>(cx, ws, p.D, ts); store_win<E>(cx, wq, p.D, tq);
            } else {
KLARA_PRAGMA_UNROLL_E
                for (int e = 0; e < E; ++e) { sm[e] = sm[e] + hf * cur.x[e]; sq[e] = sq[e] + hf * (cur.x[e] * cur.x[e]); }
            }
            held = 0;
        };
        double z[E];
        AccDraw ad = { 0.5, 0.0, false };
        const int acc_slot = (p.D + 1) >> 1;
        if (NEEDZ) lane_normals<E>(cx, p.seed, gchain, kl.t0, z, ad, acc_slot, (p.D + 1) >> 1);   // before the loaded state is touched

        TuneRegs tn;
        if (per_chain_tune) tn = { cur.step, cur.accepted, cur.proposed, cur.totproposed, 0, 0.0, 0.0 };
        else if (KPOOLED) tn = { p.tune_step[0], p.tune_accepted[0], 0, 0, 0, 0.0, 0.0 };
        else tn = { da ? cur.step : p.step0, 0, 0, 0, 0, 0.0, 0.0 };
        if (da) { tn.epsbar = cur.epsbar; tn.hbar = cur.hbar; }
        const long long acc0 = tn.accepted;
        tn.phase = per_chain_tune ? (int)(tn.proposed % p.period) : 0;
        int sphase = kl.save_phase0;
        long long scol = kl.save_col0;
        unsigned long long nacc = 0;
        bool stuck = false, last_acc = false;
        Proposal<E> prop;

        for (int s = 0; s < nsteps; ++s) {
            const unsigned long long t = kl.t0 + (unsigned long long)s;
            if (KCNT) tune_count_proposal(p, tn);
            bool acc;
            if (SAMPLER == KLARA_SAMPLER_MH) acc = step_mh<T, E, !(DIRECT || OUTER)>(p, tg, cx, gchain, t, z, ad, vp, cur.x, cur.lt, prop);
            else if (SAMPLER == KLARA_SAMPLER_MALA) acc = step_mala<T, E, PLAIN, !(DIRECT || OUTER)>(p, tg, cx, gchain, t, z, ad, tn.step, cur.x, cur.g, cur.lt, prop);
            else if (SAMPLER == KLARA_SAMPLER_HMC) {
                double a_prob = 0.0;
                acc = step_hmc<T, E, PLAIN, !(DIRECT || OUTER)>(p, tg, cx, gchain, t, z, ad, tn.step, da ? (cx.chain_ok ? da_nleaps(p, tn.step) : 1) : p.nleaps, a_prob,
                                     cur.x, cur.g, cur.lt, prop);
                if (da) da_update(p, tn, (long long)t + 1, a_prob);                   // iterate/HMC.jl:225-249
            }
            else {
                if (do_sum && held > 0) fold_state();                      // the slice sampler always moves: fold first
                acc = step_slice<T, E>(p, tg, cx, gchain, t, vp, cur.x, cur.lt, stuck);
            }
            if (OUTER) {
                if (do_sum && acc && held > 0) fold_state();               // leaving a state after `held` saved steps
                if (acc) {                                                 // commit (MH.jl:98-100, MALA.jl:95-105, HMC.jl:166-176)
KLARA_PRAGMA_UNROLL_E
                    for (int e = 0; e < E; ++e) { cur.x[e] = prop.x[e]; if (NEEDG) cur.g[e] = prop.g[e]; }
                    cur.lt = prop.lt;
                }
            }
            nacc += acc ? 1ull : 0ull;
            last_acc = acc;
            if (KCNT && acc && SAMPLER != KLARA_SAMPLER_SLICE) tn.accepted += 1;   // the slice sampler never counts accepts
            if (accept_out != nullptr && cx.chain_ok && cx.q == 0 && cx.rq == 0)
                accept_out[(long long)s * p.nchains + cx.chain] = acc ? 1 : 0;
            if (per_chain_tune && !da) tuning_block(p, tn);
            else if (per_chain_tune && tn.phase == 0 && (long long)t + 1 <= p.da_nadapt) {   // verbose report block, iterate/HMC.jl:229-243
                tn.totproposed += tn.proposed; tn.accepted = 0; tn.proposed = 0;
            }
            // save rule: BasicMCJob.jl:226-231 with postrange = (burnin+1):thinning:nsteps (BasicMCRange.jl:36)
            const long long i1 = (long long)t + 1;
            if (!NOMON && i1 > p.burnin && i1 <= p.nsteps_total) {
                if (sphase == 0) {
                    if (do_sum) held += 1;
                    // (window stores: invalid elements and the row-split's replica lanes carry out-of-range offsets — no per-element branch)
                    if (hist != nullptr && scol < p.hist_cols)
                        store_win<E>(cx, group_window(hist, scol * p.nchains + first_chain, here, p.D), p.D, cur.x);
                    if (hist_lt != nullptr && scol < p.hist_cols && cx.chain_ok && cx.q == 0 && cx.rq == 0)
                        hist_lt[scol * p.nchains + cx.chain] = cur.lt;
                    if constexpr (TARGET == KLARA_TARGET_CUSTOM) {            // :monitor => [:loglikelihood, :logprior]
                        if (p.hist_ll != nullptr && scol < p.hist_cols) {
                            double ll_, lp_;
                            tg.parts(cx, cur.x, ll_, lp_);
                            if (cx.chain_ok) { p.hist_ll[scol * p.nchains + cx.chain] = ll_; p.hist_lp[scol * p.nchains + cx.chain] = lp_; }
                        }
                    }
                    if (NEEDG && hist_g != nullptr && scol < p.hist_cols)
                        store_win<E>(cx, group_window(hist_g, scol * p.nchains + first_chain, here, p.D), p.D, cur.g);
                    ++scol;
                }
                sphase = (sphase + 1 == (int)p.thinning) ? 0 : sphase + 1;
            }
            if (!ONESTEP && NEEDZ && s + 1 < nsteps) lane_normals<E>(cx, p.seed, gchain, t + 1, z, ad, acc_slot, (p.D + 1) >> 1);
        }

        if (DIRECT) {
            if (last_acc) {
                store_win<E>(cx, group_window(p.X, first_chain, here, p.D), p.D, prop.x);
                if (NEEDG) store_win<E>(cx, group_window(p.GR, first_chain, here, p.D), p.D, prop.g);
                cur.lt = prop.lt;
            }
        } else if (nacc != 0 || SAMPLER == KLARA_SAMPLER_SLICE || nsteps > 1) {
            // (with one transition per launch a rejected proposal leaves x, g untouched: skip the write-back)
            store_win<E>(cx, group_window(p.X, first_chain, here, p.D), p.D, cur.x);
            if (NEEDG) store_win<E>(cx, group_window(p.GR, first_chain, here, p.D), p.D, cur.g);
        }
        if constexpr (!MEMSUMS) {
            if (do_sum) {
                store_win<E>(cx, group_window(p.sum, first_chain, here, p.D), p.D, sm);
                store_win<E>(cx, group_window(p.sumsq, first_chain, here, p.D), p.D, sq);
            }
        }
        if (cx.chain_ok && cx.q == 0 && cx.rq == 0) {
            if (do_sum) p.held[cx.chain] = held;
            if (nacc != 0) { p.LT[cx.chain] = cur.lt; p.naccept[cx.chain] += nacc; }
            if (da) { p.tune_step[cx.chain] = tn.step; p.da_epsbar[cx.chain] = tn.epsbar; p.da_hbar[cx.chain] = tn.hbar; }
            if (per_chain_tune) {
                p.tune_step[cx.chain] = tn.step;
                p.tune_accepted[cx.chain] = tn.accepted;
                p.tune_proposed[cx.chain] = tn.proposed;
                p.tune_totproposed[cx.chain] = tn.totproposed;
            } else if (KPOOLED && KCNT) {
                atomicAdd((unsigned long long*)p.pooled_accepted, (unsigned long long)(tn.accepted - acc0));
            }
            if (stuck) klara_raise(p.error_flag, KLARA_ERR_SLICE_STUCK);
        }
        if (!has_next) break;
        if (PREFETCH) { cur = nxt; cx = cxn; }
        else {
            set_chain<E, GT, RSPL>(p, cx, grp_next);
            load_chain<E, NEEDG, PLAIN>(p, cx, cur, grp_next * cpw, here_of(grp_next));
        }
        grp = grp_next;
    }
}

// initialize!(pstate, parameter, sampler): evaluate lt (and gradient) at X and check finiteness —
// MH.jl:72-85, MALA.jl:76-90, HMC.jl:106-120, SliceSampler.jl:40-48
template <int TARGET, int E, int GT>
__global__ __launch_bounds__(256) void k_init(const KParams p, int needgrad)
{
    kd_tables_to_lds();            // the logistic / hierarchical targets call kd_exp (table in LDS)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using T = typename TargetSel<TARGET, E>::type;
    const LaneCtx<E> cx = make_ctx<E, GT>(p);
    T tg;
    tg.init(p, cx, reinterpret_cast<double*>(smem));
    double x[E], g[E], red[1];
    load_vec<E>(cx, p.X, p.D, x);
    tg.template eval<true, true>(cx, x, red[0], g);
    group_allreduce<1>(red, cx.G, cx.lane);
    const double lt = tg.finalize(red[0]);
    bool bad = cx.chain_ok && !kfinite(lt);
    if (needgrad) {
        store_vec<E>(cx, p.GR, p.D, g);
KLARA_PRAGMA_UNROLL_E
        for (int e = 0; e < E; ++e) bad = bad || (cx.valid[e] && !kfinite(g[e]));
    }
    if (cx.chain_ok && cx.q == 0 && cx.rq == 0) p.LT[cx.chain] = lt;
    if (bad) klara_raise(p.error_flag, KLARA_ERR_NONFINITE_INIT);
}

// x0 ~ N(0, I) from the init stream (transition index 2^40-1)
template <int E, int GT>
__global__ __launch_bounds__(256) void k_init_normal(const KParams p)
{
    kd_tables_to_lds();
    const LaneCtx<E> cx = make_ctx<E, GT>(p);
    double z[E];
    AccDraw ad;
    lane_normals<E>(cx, p.seed, (unsigned long long)(p.chain_offset + cx.chain), KLARA_INIT_TRANSITION, z, ad, -1, (p.D + 1) >> 1);
    store_vec<E>(cx, p.X, p.D, z);
}
