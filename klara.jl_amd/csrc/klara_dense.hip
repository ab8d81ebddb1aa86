// klara_dense.hip — instantiates the dense-Gaussian FP64-MFMA kernels for gfx950.
#include "klara_launch.h"
#include "klara_dense.h"

// streamed layouts (NE = 40 .. 64; D = 129 .. 256; HMC, MALA, MH): klara_dense_big.hip
hipError_t klara_launch_dense_big(const KParams* p, const KLaunch& kl, int sampler, bool da, int NE, const double* Pfrag, bool hasmu, dim3 grid, hipStream_t st);
hipError_t klara_launch_dense_init_big(const KParams& p, int NE, const double* Pfrag, bool hasmu, int needgrad, dim3 grid, hipStream_t st);

template <int SAMPLER, bool DA, bool HASMU, bool PLAIN = false>
static hipError_t launch_dense_m(const KParams* p, const KLaunch& kl, int NE, const double* Pfrag, dim3 grid, hipStream_t st)
{
    const dim3 blk(512);
#define KLARA_DENSE_CASE(N)                                                                            \
    case N: {                                                                                          \
        constexpr size_t lds = sizeof(double) * (64 * (size_t)N * (size_t)((N + 3) / 4) + (HASMU ? 4 * N : 0)); \
        hipError_t e = hipFuncSetAttribute((const void*)k_dense_transitions<SAMPLER, N, DA, HASMU, PLAIN>,        \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);      \
        if (e != hipSuccess) return e;                                                                 \
        e = klara_go(k_dense_transitions<SAMPLER, N, DA, HASMU, PLAIN>, grid, blk, lds, st, p, kl, Pfrag);  \
        if (e != hipSuccess) return e;                                                                 \
        break;                                                                                         \
    }
    switch (NE) {
        KLARA_DENSE_CASE(8)
        KLARA_DENSE_CASE(16)
        KLARA_DENSE_CASE(25)
        KLARA_DENSE_CASE(32)
    default: return hipErrorInvalidValue;
    }
#undef KLARA_DENSE_CASE
    return hipGetLastError();
}

template <int SAMPLER, bool DA, bool PLAIN = false>
static hipError_t launch_dense_s(const KParams* p, const KLaunch& kl, int NE, const double* Pfrag, bool hasmu, dim3 grid, hipStream_t st)
{
    if (PLAIN) return hasmu ? launch_dense_m<SAMPLER, DA, true, PLAIN>(p, kl, NE, Pfrag, grid, st) : launch_dense_m<SAMPLER, DA, false, PLAIN>(p, kl, NE, Pfrag, grid, st);
    return hasmu ? launch_dense_m<SAMPLER, DA, true>(p, kl, NE, Pfrag, grid, st) : launch_dense_m<SAMPLER, DA, false>(p, kl, NE, Pfrag, grid, st);
}

hipError_t klara_launch_dense(const KParams* p, const KLaunch& kl, int sampler, int tuner, bool plain, int NE, const double* Pfrag, bool hasmu,
                              dim3 grid, hipStream_t st)
{
    if (NE > 32) return klara_launch_dense_big(p, kl, sampler, tuner == KLARA_TUNER_DUAL_AVERAGING, NE, Pfrag, hasmu, grid, st);
    switch (sampler) {
    case KLARA_SAMPLER_MH: return launch_dense_s<KLARA_SAMPLER_MH, false>(p, kl, NE, Pfrag, hasmu, grid, st);
    case KLARA_SAMPLER_MALA: return launch_dense_s<KLARA_SAMPLER_MALA, false>(p, kl, NE, Pfrag, hasmu, grid, st);
    case KLARA_SAMPLER_SLICE: return launch_dense_s<KLARA_SAMPLER_SLICE, false>(p, kl, NE, Pfrag, hasmu, grid, st);
    case KLARA_SAMPLER_HMC:
        if (tuner == KLARA_TUNER_DUAL_AVERAGING) return launch_dense_s<KLARA_SAMPLER_HMC, true>(p, kl, NE, Pfrag, hasmu, grid, st);
        if (plain) return launch_dense_s<KLARA_SAMPLER_HMC, false, true>(p, kl, NE, Pfrag, hasmu, grid, st);      // nothing counts or tunes: the step is a scalar, no tuner state
        return launch_dense_s<KLARA_SAMPLER_HMC, false>(p, kl, NE, Pfrag, hasmu, grid, st);
    default: return hipErrorInvalidValue;
    }
}

template <bool HASMU>
static hipError_t launch_dense_init_m(const KParams& p, int NE, const double* Pfrag, int needgrad, dim3 grid, hipStream_t st)
{
    const dim3 blk(512);
#define KLARA_DENSE_CASE(N)                                                                            \
    case N: {                                                                                          \
        constexpr size_t lds = sizeof(double) * (64 * (size_t)N * (size_t)((N + 3) / 4) + (HASMU ? 4 * N : 0)); \
        hipError_t e = hipFuncSetAttribute((const void*)k_dense_init<N, HASMU>,                        \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);      \
        if (e != hipSuccess) return e;                                                                 \
        hipLaunchKernelGGL((k_dense_init<N, HASMU>), grid, blk, lds, st, p, Pfrag, needgrad);          \
        break;                                                                                         \
    }
    switch (NE) {
        KLARA_DENSE_CASE(8)
        KLARA_DENSE_CASE(16)
        KLARA_DENSE_CASE(25)
        KLARA_DENSE_CASE(32)
    default: return hipErrorInvalidValue;
    }
#undef KLARA_DENSE_CASE
    return hipGetLastError();
}

hipError_t klara_launch_dense_init(const KParams& p, int NE, const double* Pfrag, bool hasmu, int needgrad, dim3 grid, hipStream_t st)
{
    if (NE > 32) return klara_launch_dense_init_big(p, NE, Pfrag, hasmu, needgrad, grid, st);
    return hasmu ? launch_dense_init_m<true>(p, NE, Pfrag, needgrad, grid, st) : launch_dense_init_m<false>(p, NE, Pfrag, needgrad, grid, st);
}

hipError_t klara_launch_mfma_probe(const double* A, const double* B, const double* C, double* D, hipStream_t st)
{
    hipLaunchKernelGGL(k_mfma_f64_probe, dim3(1), dim3(64), 0, st, A, B, C, D);
    return hipGetLastError();
}

hipError_t klara_launch_mfma4_probe(const double* A, const double* B, const double* C, double* D, hipStream_t st)
{
    hipLaunchKernelGGL(k_mfma_f64_4x4x4_probe, dim3(1), dim3(64), 0, st, A, B, C, D);
    return hipGetLastError();
}
