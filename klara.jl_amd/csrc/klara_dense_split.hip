// klara_dense_split.hip — instantiates the workgroup-split dense-Gaussian kernels (layout kind 6: 257 <= D <= 1024; HMC — also with dual averaging —, MALA, MH, slice) for gfx950.
#include <cstdlib>
#include "klara_launch.h"
#define KLARA_DENSE_NO_PROBES 1
#include "klara_dense_split.h"

template <int S, bool DA, bool HASMU, int MW, int NEW>
static hipError_t go_split(const KParams* p, const KLaunch& kl, int W, int D, const double* Pfrag, dim3 grid, hipStream_t st)
{
    const size_t lds = klara_split_lds_bytes(D);
    if (lds > KLARA_LDS_DEFAULT_DYNAMIC) {
        hipError_t e = hipFuncSetAttribute((const void*)k_dense_split<S, DA, HASMU, MW, NEW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    return klara_go(k_dense_split<S, DA, HASMU, MW, NEW>, grid, dim3(64 * W), lds, st, p, kl, Pfrag);
}

template <int S, bool DA, int NEW>
static hipError_t go_split_n(const KParams* p, const KLaunch& kl, int W, int D, const double* Pfrag, bool hasmu, dim3 grid, hipStream_t st)
{
    // registers for as many wavefronts per SIMD as the LDS lets a compute unit hold (160 KB, 8 KB of them the detmath tables of every workgroup):
    // W = 4: three workgroups = 3 per SIMD (168 registers); W = 8: two = 4 (128); W = 12: one = 3; W = 16: one = 4; a lone workgroup of 8, or two of 4: 2 (256)
    static const int force = getenv("KLARA_SPLIT_MW") ? atoi(getenv("KLARA_SPLIT_MW")) : 0;
    const size_t per_wg = klara_split_lds_bytes(D) + 8192;
    int nwg = (int)((size_t)160 * 1024 / per_wg); if (nwg < 1) nwg = 1;
    int mw = (nwg * W + 3) / 4; if (mw > 4) mw = 4; if (mw < 2) mw = 2;
    if (force >= 2 && force <= 4 && 4 * force >= W) mw = force;
    if (mw == 2) return hasmu ? go_split<S, DA, true, 2, NEW>(p, kl, W, D, Pfrag, grid, st) : go_split<S, DA, false, 2, NEW>(p, kl, W, D, Pfrag, grid, st);
    if (mw == 3) return hasmu ? go_split<S, DA, true, 3, NEW>(p, kl, W, D, Pfrag, grid, st) : go_split<S, DA, false, 3, NEW>(p, kl, W, D, Pfrag, grid, st);
    return hasmu ? go_split<S, DA, true, 4, NEW>(p, kl, W, D, Pfrag, grid, st) : go_split<S, DA, false, 4, NEW>(p, kl, W, D, Pfrag, grid, st);
}

template <int S, bool DA = false>
static hipError_t go_split_s(const KParams* p, const KLaunch& kl, int W, int NEW, int D, const double* Pfrag, bool hasmu, dim3 grid, hipStream_t st)
{
    if (W > KLARA_SPLIT_WMAX || W * (NEW / 4) < (D + 15) / 16) return hipErrorInvalidValue;
    if (NEW == 32) return go_split_n<S, DA, 32>(p, kl, W, D, Pfrag, hasmu, grid, st);
    if (NEW == 24) return go_split_n<S, DA, 24>(p, kl, W, D, Pfrag, hasmu, grid, st);
    if (NEW == 16) return go_split_n<S, DA, 16>(p, kl, W, D, Pfrag, hasmu, grid, st);
    return hipErrorInvalidValue;
}

hipError_t klara_launch_dense_split(const KParams* p, const KLaunch& kl, int sampler, bool da, int W, int NEW, int D, const double* Pfrag, bool hasmu, dim3 grid, hipStream_t st)
{
    switch (sampler) {
    case KLARA_SAMPLER_HMC: return da ? go_split_s<KLARA_SAMPLER_HMC, true>(p, kl, W, NEW, D, Pfrag, hasmu, grid, st) : go_split_s<KLARA_SAMPLER_HMC>(p, kl, W, NEW, D, Pfrag, hasmu, grid, st);
    case KLARA_SAMPLER_MALA: return go_split_s<KLARA_SAMPLER_MALA>(p, kl, W, NEW, D, Pfrag, hasmu, grid, st);
    case KLARA_SAMPLER_MH: return go_split_s<KLARA_SAMPLER_MH>(p, kl, W, NEW, D, Pfrag, hasmu, grid, st);
    case KLARA_SAMPLER_SLICE: return go_split_s<KLARA_SAMPLER_SLICE>(p, kl, W, NEW, D, Pfrag, hasmu, grid, st);
    default: return hipErrorInvalidValue;
    }
}

template <bool HASMU, int NEW>
static hipError_t go_split_init(const KParams& p, int W, const double* Pfrag, int needgrad, dim3 grid, hipStream_t st)
{
    const size_t lds = klara_split_lds_bytes(p.D);
    if (lds > KLARA_LDS_DEFAULT_DYNAMIC) {
        hipError_t e = hipFuncSetAttribute((const void*)k_dense_split_init<HASMU, NEW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((k_dense_split_init<HASMU, NEW>), grid, dim3(64 * W), lds, st, p, Pfrag, needgrad);
    return hipGetLastError();
}

hipError_t klara_launch_dense_split_init(const KParams& p, int W, int NEW, const double* Pfrag, bool hasmu, int needgrad, dim3 grid, hipStream_t st)
{
    if (W > KLARA_SPLIT_WMAX || W * (NEW / 4) < (p.D + 15) / 16) return hipErrorInvalidValue;
    if (NEW == 32) return hasmu ? go_split_init<true, 32>(p, W, Pfrag, needgrad, grid, st) : go_split_init<false, 32>(p, W, Pfrag, needgrad, grid, st);
    if (NEW == 24) return hasmu ? go_split_init<true, 24>(p, W, Pfrag, needgrad, grid, st) : go_split_init<false, 24>(p, W, Pfrag, needgrad, grid, st);
    if (NEW == 16) return hasmu ? go_split_init<true, 16>(p, W, Pfrag, needgrad, grid, st) : go_split_init<false, 16>(p, W, Pfrag, needgrad, grid, st);
    return hipErrorInvalidValue;
}
