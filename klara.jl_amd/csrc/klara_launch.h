// klara_launch.h — launcher prototypes implemented by the per-sampler translation units
#pragma once
#include <stdlib.h>
#include "klara_kernels.h"
#include "klara_diagt.h"

// Every transition kernel is launched through klara_go: when klara_attr_query points at a hipFuncAttributes (klara_get_kernel_attributes,
// klara_api.hip) the launcher reports the kernel's registers / scratch / LDS instead of launching it — the dispatch code that
// picks an instantiation for a job is then the one source of truth for "which kernel does this handle run".
extern thread_local hipFuncAttributes* klara_attr_query;
template <class... KArgs, class... Args>
static inline hipError_t klara_go(void (*kern)(KArgs...), dim3 grid, dim3 blk, size_t lds, hipStream_t st, Args... args)
{
    if (klara_attr_query != nullptr) return hipFuncGetAttributes(klara_attr_query, (const void*)kern);
    hipLaunchKernelGGL(kern, grid, blk, lds, st, args...);
    return hipGetLastError();
}

// group-layout transition kernels; target in {GAUSS_DIAG, LOGISTIC}; E in {2,4,8}; G = lanes per chain
hipError_t klara_launch_mh(const KParams* p, const KLaunch& kl, int mode, int target, int E, int G, dim3 grid, size_t lds,
                           hipStream_t st);
hipError_t klara_launch_mala(const KParams* p, const KLaunch& kl, int mode, int target, int E, int G, dim3 grid, size_t lds,
                           hipStream_t st);
hipError_t klara_launch_hmc(const KParams* p, const KLaunch& kl, int mode, int target, int E, int G, dim3 grid, size_t lds,
                           hipStream_t st);
hipError_t klara_launch_slice(const KParams* p, const KLaunch& kl, int mode, int target, int E, int G, dim3 grid, size_t lds,
                           hipStream_t st);
// dense (MFMA) kernels; NE in {8,16,25,32}
// (Pfrag: the fragment-ordered precision matrix, followed — hasmu — by the 4 NE zero-padded entries of the mean)
hipError_t klara_launch_dense(const KParams* p, const KLaunch& kl, int sampler, int tuner, bool plain, int NE, const double* Pfrag, bool hasmu,
                              dim3 grid, hipStream_t st);
hipError_t klara_launch_dense_init(const KParams& p, int NE, const double* Pfrag, bool hasmu, int needgrad, dim3 grid,
                                   hipStream_t st);
hipError_t klara_launch_mfma_probe(const double* A, const double* B, const double* C, double* D,
                                   hipStream_t st);
hipError_t klara_launch_mfma4_probe(const double* A, const double* B, const double* C, double* D, hipStream_t st);

// logistic regression beyond 16 parameters on the matrix cores (layout kind 5, klara_logit_mfma.h); NE in {8, 16, 24, 32}; F: the fragment stream of both
// passes in the order of consumption, ypad: the responses zero-padded to the blocks' rows (klara_api.hip logit_mfma_stream)
hipError_t klara_launch_logit_mfma(const KParams* p, const KLaunch& kl, int sampler, bool da, int NE, const double* F, const double* ypad, int nblocks,
                                   dim3 grid, hipStream_t st);
hipError_t klara_launch_logit_mfma_init(const KParams& p, int NE, const double* F, const double* ypad, int nblocks, int needgrad, dim3 grid, hipStream_t st);
int klara_logit_mfma_rbt();
// dense Gaussian on a workgroup of W = 4 ceil(ceil(D / 16) / 16) wavefronts per tile of 16 chains (layout kind 6, klara_dense_split.h): 257 <= D <= 1024; MH, MALA, HMC
// elements per lane and wavefront of the split dense layout (4 per row tile of P a wavefront can own): 16, 24 or 32 — whichever puts the FEWEST wavefronts on a
// tile (the smaller one on a tie): the fewest wavefronts that hold the tile measured fastest at every size (profiles/r6_dense_split.txt), as long as their
// registers are not short — so 257 <= D <= 384: 24 (4 wavefronts), 385 .. 512: 32 (4), 513 .. 768: 24 (8), 769 .. 1024: 32 (8; 24 there would be 12 wavefronts at
// 168 registers with scratch: measured slower than 16).  KLARA_SPLIT_NEW=16|24|32 forces one.
static inline int klara_split_new(int D)
{
    const int MT = (D + 15) / 16;
    if (const char* e = getenv("KLARA_SPLIT_NEW")) { const int v = atoi(e); if (v == 16 || v == 24 || v == 32) return v; }
    int best = 16, wbest = 4 * ((MT + 15) / 16);
    for (int n = 24; n <= 32; n += 8) { const int w = 4 * ((MT + n - 1) / n); if (w < wbest && w <= 8) { best = n; wbest = w; } }
    return best;
}
// wavefronts per tile of 16 chains: whole SIMD rounds, at most klara_split_new / 4 row tiles of P per wavefront
static inline int klara_split_waves(int D)
{
    const int MT = (D + 15) / 16, n = klara_split_new(D), w = 4 * ((MT + n - 1) / n);
    if (const char* e = getenv("KLARA_SPLIT_W")) { const int v = atoi(e); if (v >= w && v <= 16 && v % 4 == 0) return v; }      // (measurements: more wavefronts, fewer tiles each)
    return w;
}
// LDS bytes of a workgroup: xb (4 MT rows of 64 doubles) + the partial sums (the 8 KB of detmath tables are static)
static inline size_t klara_split_lds_bytes(int D)
{
    const size_t MT = ((size_t)D + 15) / 16;
    return sizeof(double) * (4 * MT * 64 + 2 * 3 * (size_t)klara_split_waves(D) * 16);
}
hipError_t klara_launch_dense_split(const KParams* p, const KLaunch& kl, int sampler, bool da, int W, int NEW, int D, const double* Pfrag, bool hasmu, dim3 grid, hipStream_t st);
hipError_t klara_launch_dense_split_init(const KParams& p, int W, int NEW, const double* Pfrag, bool hasmu, int needgrad, dim3 grid, hipStream_t st);      // row tiles per block the kernels were built for

// pair-transposed diagonal-Gaussian kernels (layout kind 3, klara_diagt.h).  The translation units klara_diagt_*.hip are
// compiled four times: Q = 8 lanes per chain (17 <= D <= 128, NP = ceil(D/16) in 2..8), Q = 16 (129 <= D <= 256), Q = 32
// (257 <= D <= 512) and (round 6) Q = 64 (513 <= D <= 1024: north_star's one chain per wavefront), the latter three with NP in 5..8; the launchers of the
// wider variants carry a _q16 / _q32 / _q64 suffix.
#if KLARA_DIAGT_Q == 8
#define KLARA_DIAGT_FN(name) name
#elif KLARA_DIAGT_Q == 4
#define KLARA_DIAGT_FN(name) name##_q4
#elif KLARA_DIAGT_Q == 16
#define KLARA_DIAGT_FN(name) name##_q16
#elif KLARA_DIAGT_Q == 32
#define KLARA_DIAGT_FN(name) name##_q32
#elif KLARA_DIAGT_Q == 64
#define KLARA_DIAGT_FN(name) name##_q64
#else
#define KLARA_DIAGT_FN(name) name          // (experimental lane counts replace the Q = 8 set)
#endif
#define KLARA_DIAGT_DECLARE(SUFFIX)                                                                                                                       \
    hipError_t klara_launch_diagt_mh##SUFFIX(const KParams* p, const KLaunch& kl, int NP, bool onestep, bool unitw, bool mon, bool tune, bool da, const KAuto& ka, long long nwaves, hipStream_t st);   \
    hipError_t klara_launch_diagt_mala##SUFFIX(const KParams* p, const KLaunch& kl, int NP, bool onestep, bool unitw, bool mon, bool tune, bool da, const KAuto& ka, long long nwaves, hipStream_t st); \
    hipError_t klara_launch_diagt_hmc##SUFFIX(const KParams* p, const KLaunch& kl, int NP, bool onestep, bool unitw, bool mon, bool tune, bool da, const KAuto& ka, long long nwaves, hipStream_t st);  \
    hipError_t klara_launch_diagt_slice##SUFFIX(const KParams* p, const KLaunch& kl, int NP, bool unitw, bool mon, bool tune, const KAuto& ka, long long nwaves, hipStream_t st);                      \
    hipError_t klara_launch_diagt_slice_free##SUFFIX(const KParams* p, const KLaunch& kl, int NP, bool unitw, bool mon, const KAuto& ka, long long nwaves, hipStream_t st);                           \
    hipError_t klara_launch_diagt_hist_lt##SUFFIX(const KParams* p, const KLaunch& kl, int NP, bool unitw, long long col0, int ncols, long long ngroups, hipStream_t st);                              \
    hipError_t klara_launch_diagt_init##SUFFIX(const KParams& p, int NP, int needgrad, dim3 grid, hipStream_t st);
KLARA_DIAGT_DECLARE()
KLARA_DIAGT_DECLARE(_q16)
KLARA_DIAGT_DECLARE(_q32)
KLARA_DIAGT_DECLARE(_q64)
// Q = 4 lanes per chain, 16 chains per wavefront, NP = ceil(D/8) in 3..13 (17 <= D <= 104): jobs in which nothing counts or tunes
// (VanillaMCTuner, not verbose) with the MH or the MALA sampler — D = 100 occupies 50 of 52 pair slots instead of 50 of 56, the
// per-wavefront work (reductions, accept test, addressing) is shared by 16 chains, and the running sums are folded with atomic
// adds instead of living in registers (klara_diagt.h diagt_fold_atomic).  Only klara_diagt_{mh,mala,init}.hip are built for it.
hipError_t klara_launch_diagt_mh_q4(const KParams* p, const KLaunch& kl, int NP, bool onestep, bool unitw, bool mon, bool tune, bool da, const KAuto& ka, long long nwaves, hipStream_t st);
hipError_t klara_launch_diagt_mala_q4(const KParams* p, const KLaunch& kl, int NP, bool onestep, bool unitw, bool mon, bool tune, bool da, const KAuto& ka, long long nwaves, hipStream_t st);
hipError_t klara_launch_diagt_hmc_q4(const KParams* p, const KLaunch& kl, int NP, bool onestep, bool unitw, bool mon, bool tune, bool da, const KAuto& ka, long long nwaves, hipStream_t st);
hipError_t klara_launch_diagt_init_q4(const KParams& p, int NP, int needgrad, dim3 grid, hipStream_t st);
// pairs per lane the kernels are instantiated for; a job takes NP = ceil(ceil(D/2) / Q) exactly (only the LAST pair of a lane
// can be padding)
#if KLARA_DIAGT_Q == 4
#define KLARA_DIAGT_NP_MENU_DO(X) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13)
#elif KLARA_DIAGT_Q == 8
#define KLARA_DIAGT_NP_MENU_DO(X) X(2) X(3) X(4) X(5) X(6) X(7) X(8)
#else                      // Q = 16 / 32 start where the narrower variant ends: NP = 5..8
#define KLARA_DIAGT_NP_MENU_DO(X) X(5) X(6) X(7) X(8)
#endif
#define KLARA_DIAGT_NP_MAX 8

// one launch of a pair-transposed kernel over `nwaves` chain groups: one wavefront each, four per workgroup
template <int S, int NP_, int Q_, bool ONESTEP, bool UNITW, bool MON, bool TUNE = false, bool DA = false>
static hipError_t diagt_go(const KParams* p, const KLaunch& kl, const KAuto& ka, long long nwaves, hipStream_t st)
{
    return klara_go(k_diagt<S, NP_, Q_, ONESTEP, UNITW, MON, TUNE, DA>, dim3((unsigned)((nwaves + 3) / 4)), dim3(256), 0, st, p, kl, ka);
}

#if KLARA_DIAGT_Q == 4     // (no tuned / dual-averaging instantiations: those jobs take the 8-lane form)
#define KLARA_DIAGT_CASE(S, NP_)                                                                                   \
    case NP_:                                                                                                      \
        if (tune || da) return hipErrorInvalidValue;                                                               \
        else if (mon && unitw) e_ = diagt_go<S, NP_, KLARA_DIAGT_Q, false, true, true>(p, kl, ka, nwaves, st); \
        else if (mon) e_ = diagt_go<S, NP_, KLARA_DIAGT_Q, false, false, true>(p, kl, ka, nwaves, st);        \
        else if (onestep && unitw) e_ = diagt_go<S, NP_, KLARA_DIAGT_Q, true, true, false>(p, kl, ka, nwaves, st);  \
        else if (onestep) e_ = diagt_go<S, NP_, KLARA_DIAGT_Q, true, false, false>(p, kl, ka, nwaves, st);    \
        else if (unitw) e_ = diagt_go<S, NP_, KLARA_DIAGT_Q, false, true, false>(p, kl, ka, nwaves, st);      \
        else e_ = diagt_go<S, NP_, KLARA_DIAGT_Q, false, false, false>(p, kl, ka, nwaves, st);                \
        break;
#else
#define KLARA_DIAGT_CASE(S, NP_)                                                                                   \
    case NP_:                                                                                                      \
        if (da && unitw) e_ = diagt_go<S, NP_, KLARA_DIAGT_Q, false, true, true, true, (S == KLARA_SAMPLER_HMC)>(p, kl, ka, nwaves, st); \
        else if (da) e_ = diagt_go<S, NP_, KLARA_DIAGT_Q, false, false, true, true, (S == KLARA_SAMPLER_HMC)>(p, kl, ka, nwaves, st); \
        else if (tune && unitw) e_ = diagt_go<S, NP_, KLARA_DIAGT_Q, false, true, true, true>(p, kl, ka, nwaves, st);  \
        else if (tune) e_ = diagt_go<S, NP_, KLARA_DIAGT_Q, false, false, true, true>(p, kl, ka, nwaves, st);  \
        else if (mon && unitw) e_ = diagt_go<S, NP_, KLARA_DIAGT_Q, false, true, true>(p, kl, ka, nwaves, st); \
        else if (mon) e_ = diagt_go<S, NP_, KLARA_DIAGT_Q, false, false, true>(p, kl, ka, nwaves, st);        \
        else if (onestep && unitw) e_ = diagt_go<S, NP_, KLARA_DIAGT_Q, true, true, false>(p, kl, ka, nwaves, st);  \
        else if (onestep) e_ = diagt_go<S, NP_, KLARA_DIAGT_Q, true, false, false>(p, kl, ka, nwaves, st);    \
        else if (unitw) e_ = diagt_go<S, NP_, KLARA_DIAGT_Q, false, true, false>(p, kl, ka, nwaves, st);      \
        else e_ = diagt_go<S, NP_, KLARA_DIAGT_Q, false, false, false>(p, kl, ka, nwaves, st);                \
        break;
#endif
#define KLARA_DIAGT_CASE_KLARA_SAMPLER_MH(NP_) KLARA_DIAGT_CASE(KLARA_SAMPLER_MH, NP_)
#define KLARA_DIAGT_CASE_KLARA_SAMPLER_MALA(NP_) KLARA_DIAGT_CASE(KLARA_SAMPLER_MALA, NP_)
#define KLARA_DIAGT_CASE_KLARA_SAMPLER_HMC(NP_) KLARA_DIAGT_CASE(KLARA_SAMPLER_HMC, NP_)
#define KLARA_DISPATCH_DIAGT(S)                                                                                    \
    do {                                                                                                           \
        hipError_t e_ = hipSuccess;                                                                                \
        switch (NP) {                                                                                              \
            KLARA_DIAGT_NP_MENU_DO(KLARA_DIAGT_CASE_##S)                                                           \
            default: return hipErrorInvalidValue;                                                                  \
        }                                                                                                          \
        return e_;                                                                                                 \
    } while (0)

// MH / MALA / HMC on the hierarchical target, 8 lanes per chain (layout kind 4, klara_hiert.h); RPL = 4 units per lane, NT = 5
hipError_t klara_launch_hiert(const KParams* p, const KLaunch& kl, int sampler, int RPL, int NT, bool mon, bool tune, bool da, dim3 grid,
                              hipStream_t st);
hipError_t klara_launch_hiert_init(const KParams& p, int RPL, int NT, int needgrad, dim3 grid, hipStream_t st);

// user-defined targets (KLARA_TARGET_CUSTOM): run-time compiled instantiations of k_init / k_transitions (klara_jit.hip);
// `modes` are the k_transitions MODE values the job can launch; load = false only compiles (no GPU needed)
struct KlaraJit;
klara_status klara_jit_create(const char* src, int sampler, int D, int E, int G, const int* modes, int nmodes, bool load, KlaraJit** out);
// pair closures (`#define KLARA_USER_PAIR_TARGET 1` + klara_user_pair, klara_diagt.h USERPAIR): k_diagt_init / k_diagt instantiated
// for the job's NP pairs per lane, Q lanes per chain and monitor / tuner flags; modes: 0 = fused launches, 1 = one transition per launch
klara_status klara_jit_create_pair(const char* src, int sampler, int D, int NP, int Q, bool mon, bool tune, bool da, const int* modes, int nmodes,
                                   bool load, KlaraJit** out);
hipError_t klara_jit_launch_pair(KlaraJit* j, int mode, const KParams* p, const KLaunch& kl, long long nwaves, hipStream_t st);
void klara_jit_destroy(KlaraJit* j);
hipError_t klara_jit_launch_init(KlaraJit* j, const KParams& p, int needgrad, dim3 grid, size_t lds, hipStream_t st, int block = 256);
hipError_t klara_jit_launch(KlaraJit* j, int mode, const KParams* p, const KLaunch& kl, dim3 grid, size_t lds, hipStream_t st, int block = 256);
const char* klara_jit_log();

// mode 7: mode 3 with exactly one transition per launch; mode 3: nothing counts/tunes and nothing is monitored;
// mode 1: nothing counts/tunes; mode 0: general
// (a launch gets 64 KB of LDS — 8 KB of math tables + 56 KB of dynamic — without asking; the logistic target's data rows may need more)
#ifndef KLARA_LDS_DEFAULT_DYNAMIC
#define KLARA_LDS_DEFAULT_DYNAMIC 57344u
#endif
#define KLARA_LAUNCH_TM(S, T, E_, G_, M_)                                                                                          \
    do {                                                                                                                          \
        if (lds > KLARA_LDS_DEFAULT_DYNAMIC) {                                                                                    \
            hipError_t e_ = hipFuncSetAttribute((const void*)k_transitions<S, T, E_, G_, M_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e_ != hipSuccess) return e_;                                                                                      \
        }                                                                                                                         \
        { const hipError_t e_ = klara_go(k_transitions<S, T, E_, G_, M_>, grid, blk, lds, st, p, kl); if (e_ != hipSuccess) return e_; }  \
    } while (0)
#define KLARA_LAUNCH_T(S, T, E_, G_)                                                                    \
    do {                                                                                               \
        if (mode == 7) KLARA_LAUNCH_TM(S, T, E_, G_, 7);                                                \
        else if ((mode & 3) == 3) KLARA_LAUNCH_TM(S, T, E_, G_, 3);                                     \
        else if (mode & 1) KLARA_LAUNCH_TM(S, T, E_, G_, 1);                                            \
        else KLARA_LAUNCH_TM(S, T, E_, G_, 0);                                                          \
    } while (0)

// dispatch helper used by every group-layout launcher
#define KLARA_DISPATCH_GROUP(KERNEL_EXPR_PREFIX, SAMPLER)                                              \
    do {                                                                                               \
        const dim3 blk(256);                                                                           \
        if (target == KLARA_TARGET_GAUSS_DIAG) {                                                       \
            if (E == 2 && G == 64) KLARA_LAUNCH_T(SAMPLER, KLARA_TARGET_GAUSS_DIAG, 2, 64); \
            else if (E == 2) KLARA_LAUNCH_T(SAMPLER, KLARA_TARGET_GAUSS_DIAG, 2, 0);       \
            else if (E == 4 && G == 32) KLARA_LAUNCH_T(SAMPLER, KLARA_TARGET_GAUSS_DIAG, 4, 32); \
            else if (E == 4) KLARA_LAUNCH_T(SAMPLER, KLARA_TARGET_GAUSS_DIAG, 4, 0);       \
            else if (E == 8) KLARA_LAUNCH_T(SAMPLER, KLARA_TARGET_GAUSS_DIAG, 8, 0);       \
            else return hipErrorInvalidValue;                                                          \
        } else if (target == KLARA_TARGET_LOGISTIC) {                                                  \
            if (E == 2) KLARA_LAUNCH_T(SAMPLER, KLARA_TARGET_LOGISTIC, 2, 0);              \
            else if (E == 4) KLARA_LAUNCH_T(SAMPLER, KLARA_TARGET_LOGISTIC, 4, 0);         \
            else if (E == 8) KLARA_LAUNCH_T(SAMPLER, KLARA_TARGET_LOGISTIC, 8, 0);         \
            else if (E == 16) KLARA_LAUNCH_T(SAMPLER, KLARA_TARGET_LOGISTIC, 16, 0);       \
            else return hipErrorInvalidValue;                                                          \
        } else if (target == KLARA_TARGET_HIER_NORMAL) {                                               \
            if (E == 2) KLARA_LAUNCH_T(SAMPLER, KLARA_TARGET_HIER_NORMAL, 2, 0);           \
            else if (E == 4) KLARA_LAUNCH_T(SAMPLER, KLARA_TARGET_HIER_NORMAL, 4, 0);      \
            else return hipErrorInvalidValue;                                                          \
        } else return hipErrorInvalidValue;                                                            \
        return hipGetLastError();                                                                      \
    } while (0)
