// klara_dense_split.hip — instantiates the workgroup-split dense-Gaussian kernels (layout kind 6: 257 <= D <= 1024; HMC — also with dual averaging —, MALA, MH, slice) for gfx950.
#include <cstdlib>
#include "klara_launch.h"
#define KLARA_DENSE_NO_PROBES 1
#include "klara_dense_split.h"

template <int S, bool DA, bool HASMU, int MW>
static hipError_t go_split(const KParams* p, const KLaunch& kl, int W, int D, const double* Pfrag, dim3 grid, hipStream_t st)
{
    const size_t lds = klara_split_lds_bytes(D);
    if (lds > KLARA_LDS_DEFAULT_DYNAMIC) {
        hipError_t e = hipFuncSetAttribute((const void*)k_dense_split<S, DA, HASMU, MW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    return klara_go(k_dense_split<S, DA, HASMU, MW>, grid, dim3(64 * W), lds, st, p, kl, Pfrag);
}

template <int S, bool DA = false>
static hipError_t go_split_s(const KParams* p, const KLaunch& kl, int W, int D, const double* Pfrag, bool hasmu, dim3 grid, hipStream_t st)
{
    if (W != klara_split_waves(D) || W > KLARA_SPLIT_WMAX) return hipErrorInvalidValue;
    // registers for 3 wavefronts per SIMD where the LDS or the workgroup's size admit no more (W = 4: three workgroups per compute unit; W = 12: one),
    // for 4 otherwise (W = 8: two workgroups; W = 16: one)
    static const int force = getenv("KLARA_SPLIT_MW") ? atoi(getenv("KLARA_SPLIT_MW")) : 0;
    if (force == 3 ? W <= 12 : (force != 4 && (W == 4 || W == 12))) return hasmu ? go_split<S, DA, true, 3>(p, kl, W, D, Pfrag, grid, st) : go_split<S, DA, false, 3>(p, kl, W, D, Pfrag, grid, st);
    return hasmu ? go_split<S, DA, true, 4>(p, kl, W, D, Pfrag, grid, st) : go_split<S, DA, false, 4>(p, kl, W, D, Pfrag, grid, st);
}

hipError_t klara_launch_dense_split(const KParams* p, const KLaunch& kl, int sampler, bool da, int W, int D, const double* Pfrag, bool hasmu, dim3 grid, hipStream_t st)
{
    switch (sampler) {
    case KLARA_SAMPLER_HMC: return da ? go_split_s<KLARA_SAMPLER_HMC, true>(p, kl, W, D, Pfrag, hasmu, grid, st) : go_split_s<KLARA_SAMPLER_HMC>(p, kl, W, D, Pfrag, hasmu, grid, st);
    case KLARA_SAMPLER_MALA: return go_split_s<KLARA_SAMPLER_MALA>(p, kl, W, D, Pfrag, hasmu, grid, st);
    case KLARA_SAMPLER_MH: return go_split_s<KLARA_SAMPLER_MH>(p, kl, W, D, Pfrag, hasmu, grid, st);
    case KLARA_SAMPLER_SLICE: return go_split_s<KLARA_SAMPLER_SLICE>(p, kl, W, D, Pfrag, hasmu, grid, st);
    default: return hipErrorInvalidValue;
    }
}

hipError_t klara_launch_dense_split_init(const KParams& p, int W, const double* Pfrag, bool hasmu, int needgrad, dim3 grid, hipStream_t st)
{
    if (W != klara_split_waves(p.D) || W > KLARA_SPLIT_WMAX) return hipErrorInvalidValue;
    const size_t lds = klara_split_lds_bytes(p.D);
    const void* fn = hasmu ? (const void*)k_dense_split_init<true> : (const void*)k_dense_split_init<false>;
    if (lds > KLARA_LDS_DEFAULT_DYNAMIC) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    if (hasmu) hipLaunchKernelGGL((k_dense_split_init<true>), grid, dim3(64 * W), lds, st, p, Pfrag, needgrad);
    else hipLaunchKernelGGL((k_dense_split_init<false>), grid, dim3(64 * W), lds, st, p, Pfrag, needgrad);
    return hipGetLastError();
}
