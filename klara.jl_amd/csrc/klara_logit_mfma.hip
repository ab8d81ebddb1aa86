// klara_logit_mfma.hip — instantiates the matrix-core logistic-regression kernels (layout kind 5: 17 <= D <= 256, NE = 8, 16, 24, 32, 40, 48, 56, 64 elements per lane;
// MH, MALA, HMC — also with dual averaging —, slice) for gfx950.
#include "klara_launch.h"
#define KLARA_DENSE_NO_PROBES 1
#include "klara_logit_mfma.h"

template <int S, int N, bool DA = false>
static hipError_t go_logitm(const KParams* p, const KLaunch& kl, const double* F, const double* ypad, int nblocks, dim3 grid, hipStream_t st)
{
    // MH's sigma + one column per lane of the four wavefronts (momentum / normals / current value)
    constexpr size_t lds = sizeof(double) * (256 + (S == KLARA_SAMPLER_MH ? 4 * N : 0) + 4 * (size_t)N * 64);      // + kd_log12's table
    if (lds > KLARA_LDS_DEFAULT_DYNAMIC) {
        hipError_t e = hipFuncSetAttribute((const void*)k_logit_mfma<S, N, DA>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    return klara_go(k_logit_mfma<S, N, DA>, grid, dim3(256), lds, st, p, kl, F, ypad, nblocks);
}

template <int S, bool DA = false>
static hipError_t go_logitm_s(const KParams* p, const KLaunch& kl, int NE, const double* F, const double* ypad, int nblocks, dim3 grid, hipStream_t st)
{
    if (NE == 8) return go_logitm<S, 8, DA>(p, kl, F, ypad, nblocks, grid, st);
    if (NE == 16) return go_logitm<S, 16, DA>(p, kl, F, ypad, nblocks, grid, st);
    if (NE == 24) return go_logitm<S, 24, DA>(p, kl, F, ypad, nblocks, grid, st);
    if (NE == 32) return go_logitm<S, 32, DA>(p, kl, F, ypad, nblocks, grid, st);
    if (NE == 40) return go_logitm<S, 40, DA>(p, kl, F, ypad, nblocks, grid, st);
    if (NE == 48) return go_logitm<S, 48, DA>(p, kl, F, ypad, nblocks, grid, st);
    if (NE == 56) return go_logitm<S, 56, DA>(p, kl, F, ypad, nblocks, grid, st);
    if (NE == 64) return go_logitm<S, 64, DA>(p, kl, F, ypad, nblocks, grid, st);
    return hipErrorInvalidValue;
}

hipError_t klara_launch_logit_mfma(const KParams* p, const KLaunch& kl, int sampler, bool da, int NE, const double* F, const double* ypad, int nblocks, dim3 grid, hipStream_t st)
{
    switch (sampler) {
    case KLARA_SAMPLER_HMC: return da ? go_logitm_s<KLARA_SAMPLER_HMC, true>(p, kl, NE, F, ypad, nblocks, grid, st) : go_logitm_s<KLARA_SAMPLER_HMC>(p, kl, NE, F, ypad, nblocks, grid, st);
    case KLARA_SAMPLER_MALA: return go_logitm_s<KLARA_SAMPLER_MALA>(p, kl, NE, F, ypad, nblocks, grid, st);
    case KLARA_SAMPLER_MH: return go_logitm_s<KLARA_SAMPLER_MH>(p, kl, NE, F, ypad, nblocks, grid, st);
    case KLARA_SAMPLER_SLICE: return go_logitm_s<KLARA_SAMPLER_SLICE>(p, kl, NE, F, ypad, nblocks, grid, st);
    default: return hipErrorInvalidValue;
    }
}

hipError_t klara_launch_logit_mfma_init(const KParams& p, int NE, const double* F, const double* ypad, int nblocks, int needgrad, dim3 grid, hipStream_t st)
{
    if (NE == 8) hipLaunchKernelGGL((k_logit_mfma_init<8>), grid, dim3(256), 0, st, p, F, ypad, nblocks, needgrad);
    else if (NE == 16) hipLaunchKernelGGL((k_logit_mfma_init<16>), grid, dim3(256), 0, st, p, F, ypad, nblocks, needgrad);
    else if (NE == 24) hipLaunchKernelGGL((k_logit_mfma_init<24>), grid, dim3(256), 0, st, p, F, ypad, nblocks, needgrad);
    else if (NE == 32) hipLaunchKernelGGL((k_logit_mfma_init<32>), grid, dim3(256), 0, st, p, F, ypad, nblocks, needgrad);
    else if (NE == 40) hipLaunchKernelGGL((k_logit_mfma_init<40>), grid, dim3(256), 0, st, p, F, ypad, nblocks, needgrad);
    else if (NE == 48) hipLaunchKernelGGL((k_logit_mfma_init<48>), grid, dim3(256), 0, st, p, F, ypad, nblocks, needgrad);
    else if (NE == 56) hipLaunchKernelGGL((k_logit_mfma_init<56>), grid, dim3(256), 0, st, p, F, ypad, nblocks, needgrad);
    else if (NE == 64) hipLaunchKernelGGL((k_logit_mfma_init<64>), grid, dim3(256), 0, st, p, F, ypad, nblocks, needgrad);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

int klara_logit_mfma_rbt() { return KLARA_LOGITM_RBT; }
