// klara_dense_split.hip — instantiates the workgroup-split dense-Gaussian kernels (layout kind 6: 257 <= D <= 1024; HMC — also with dual averaging —, MALA, MH) for gfx950.
#include "klara_launch.h"
#define KLARA_DENSE_NO_PROBES 1
#include "klara_dense_split.h"

template <int S, bool DA, bool HASMU, int WB>
static hipError_t go_split(const KParams* p, const KLaunch& kl, int W, const double* Pfrag, dim3 grid, hipStream_t st)
{
    const size_t lds = klara_split_lds_bytes(W, HASMU);
    if (lds > KLARA_LDS_DEFAULT_DYNAMIC) {
        hipError_t e = hipFuncSetAttribute((const void*)k_dense_split<S, DA, HASMU, WB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    return klara_go(k_dense_split<S, DA, HASMU, WB>, grid, dim3(64 * W), lds, st, p, kl, Pfrag);
}

template <int S, bool DA = false>
static hipError_t go_split_s(const KParams* p, const KLaunch& kl, int W, const double* Pfrag, bool hasmu, dim3 grid, hipStream_t st)
{
    if (W < 1 || W > KLARA_SPLIT_WMAX) return hipErrorInvalidValue;
    if (W <= 4) return hasmu ? go_split<S, DA, true, 4>(p, kl, W, Pfrag, grid, st) : go_split<S, DA, false, 4>(p, kl, W, Pfrag, grid, st);
    if (W <= 8) return hasmu ? go_split<S, DA, true, 8>(p, kl, W, Pfrag, grid, st) : go_split<S, DA, false, 8>(p, kl, W, Pfrag, grid, st);
    return hasmu ? go_split<S, DA, true, 16>(p, kl, W, Pfrag, grid, st) : go_split<S, DA, false, 16>(p, kl, W, Pfrag, grid, st);
}

hipError_t klara_launch_dense_split(const KParams* p, const KLaunch& kl, int sampler, bool da, int W, const double* Pfrag, bool hasmu, dim3 grid, hipStream_t st)
{
    switch (sampler) {
    case KLARA_SAMPLER_HMC: return da ? go_split_s<KLARA_SAMPLER_HMC, true>(p, kl, W, Pfrag, hasmu, grid, st) : go_split_s<KLARA_SAMPLER_HMC>(p, kl, W, Pfrag, hasmu, grid, st);
    case KLARA_SAMPLER_MALA: return go_split_s<KLARA_SAMPLER_MALA>(p, kl, W, Pfrag, hasmu, grid, st);
    case KLARA_SAMPLER_MH: return go_split_s<KLARA_SAMPLER_MH>(p, kl, W, Pfrag, hasmu, grid, st);
    default: return hipErrorInvalidValue;
    }
}

hipError_t klara_launch_dense_split_init(const KParams& p, int W, const double* Pfrag, bool hasmu, int needgrad, dim3 grid, hipStream_t st)
{
    if (W < 1 || W > KLARA_SPLIT_WMAX) return hipErrorInvalidValue;
    const size_t lds = klara_split_lds_bytes(W, hasmu);
    const void* fn = hasmu ? (const void*)k_dense_split_init<true> : (const void*)k_dense_split_init<false>;
    if (lds > KLARA_LDS_DEFAULT_DYNAMIC) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    if (hasmu) hipLaunchKernelGGL((k_dense_split_init<true>), grid, dim3(64 * W), lds, st, p, Pfrag, needgrad);
    else hipLaunchKernelGGL((k_dense_split_init<false>), grid, dim3(64 * W), lds, st, p, Pfrag, needgrad);
    return hipGetLastError();
}
