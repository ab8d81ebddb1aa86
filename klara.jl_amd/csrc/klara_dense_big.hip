// klara_dense_big.hip — instantiates the streamed dense-Gaussian kernels (D = 129 .. 256: NE = 40, 48, 56, 64 elements per lane; HMC — also with dual averaging —, MALA, MH, slice) for gfx950.
#include "klara_launch.h"
#define KLARA_DENSE_NO_PROBES 1
#include "klara_dense_big.h"

template <int S, int N, bool HASMU, bool DA = false>
static hipError_t go_big(const KParams* p, const KLaunch& kl, const double* Pfrag, dim3 grid, hipStream_t st)
{
    // mu + (HMC: momentum, MALA: the proposal's normals) the four wavefronts' columns
    constexpr size_t lds = sizeof(double) * ((HASMU ? 4 * N : 0) + (S == KLARA_SAMPLER_MH ? 4 * N : 0) + (S == KLARA_SAMPLER_SLICE ? 0 : 4 * (size_t)N * 64));      // mu, MH's sigma, one column per lane of the 4 wavefronts (none for the slice sampler)
    if (lds > KLARA_LDS_DEFAULT_DYNAMIC) {
        hipError_t e = hipFuncSetAttribute((const void*)k_dense_big<S, N, HASMU, DA>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    return klara_go(k_dense_big<S, N, HASMU, DA>, grid, dim3(256), lds, st, p, kl, Pfrag);
}

template <int S, bool DA = false>
static hipError_t go_big_s(const KParams* p, const KLaunch& kl, int NE, const double* Pfrag, bool hasmu, dim3 grid, hipStream_t st)
{
    if (NE == 40) return hasmu ? go_big<S, 40, true, DA>(p, kl, Pfrag, grid, st) : go_big<S, 40, false, DA>(p, kl, Pfrag, grid, st);
    if (NE == 48) return hasmu ? go_big<S, 48, true, DA>(p, kl, Pfrag, grid, st) : go_big<S, 48, false, DA>(p, kl, Pfrag, grid, st);
    if (NE == 56) return hasmu ? go_big<S, 56, true, DA>(p, kl, Pfrag, grid, st) : go_big<S, 56, false, DA>(p, kl, Pfrag, grid, st);
    if (NE == 64) return hasmu ? go_big<S, 64, true, DA>(p, kl, Pfrag, grid, st) : go_big<S, 64, false, DA>(p, kl, Pfrag, grid, st);
    return hipErrorInvalidValue;
}

hipError_t klara_launch_dense_big(const KParams* p, const KLaunch& kl, int sampler, bool da, int NE, const double* Pfrag, bool hasmu, dim3 grid, hipStream_t st)
{
    switch (sampler) {
    case KLARA_SAMPLER_HMC: return da ? go_big_s<KLARA_SAMPLER_HMC, true>(p, kl, NE, Pfrag, hasmu, grid, st) : go_big_s<KLARA_SAMPLER_HMC>(p, kl, NE, Pfrag, hasmu, grid, st);
    case KLARA_SAMPLER_MALA: return go_big_s<KLARA_SAMPLER_MALA>(p, kl, NE, Pfrag, hasmu, grid, st);
    case KLARA_SAMPLER_MH: return go_big_s<KLARA_SAMPLER_MH>(p, kl, NE, Pfrag, hasmu, grid, st);
    case KLARA_SAMPLER_SLICE: return go_big_s<KLARA_SAMPLER_SLICE>(p, kl, NE, Pfrag, hasmu, grid, st);
    default: return hipErrorInvalidValue;
    }
}

template <int N, bool HASMU>
static hipError_t go_init_big(const KParams& p, const double* Pfrag, int needgrad, dim3 grid, hipStream_t st)
{
    hipLaunchKernelGGL((k_dense_init_big<N, HASMU>), grid, dim3(256), sizeof(double) * (HASMU ? 4 * N : 0), st, p, Pfrag, needgrad);
    return hipGetLastError();
}

hipError_t klara_launch_dense_init_big(const KParams& p, int NE, const double* Pfrag, bool hasmu, int needgrad, dim3 grid, hipStream_t st)
{
    if (NE == 40) return hasmu ? go_init_big<40, true>(p, Pfrag, needgrad, grid, st) : go_init_big<40, false>(p, Pfrag, needgrad, grid, st);
    if (NE == 48) return hasmu ? go_init_big<48, true>(p, Pfrag, needgrad, grid, st) : go_init_big<48, false>(p, Pfrag, needgrad, grid, st);
    if (NE == 56) return hasmu ? go_init_big<56, true>(p, Pfrag, needgrad, grid, st) : go_init_big<56, false>(p, Pfrag, needgrad, grid, st);
    if (NE == 64) return hasmu ? go_init_big<64, true>(p, Pfrag, needgrad, grid, st) : go_init_big<64, false>(p, Pfrag, needgrad, grid, st);
    return hipErrorInvalidValue;
}
