// klara_diagt_slice.hip — instantiates the pair-transposed slice-sampler kernels (layout kind 3) for gfx950.
#include "klara_launch.h"

#define KLARA_DIAGT_SLICE_CASE(NP_)                                                                                       \
    case NP_:                                                                                                              \
        if (tune && unitw) hipLaunchKernelGGL((k_diagt<KLARA_SAMPLER_SLICE, NP_, KLARA_DIAGT_Q, false, true, true, true>), grid, blk, 0, st, p, kl);        \
        else if (tune) hipLaunchKernelGGL((k_diagt<KLARA_SAMPLER_SLICE, NP_, KLARA_DIAGT_Q, false, false, true, true>), grid, blk, 0, st, p, kl);       \
        else if (mon && unitw) hipLaunchKernelGGL((k_diagt<KLARA_SAMPLER_SLICE, NP_, KLARA_DIAGT_Q, false, true, true>), grid, blk, 0, st, p, kl);      \
        else if (mon) hipLaunchKernelGGL((k_diagt<KLARA_SAMPLER_SLICE, NP_, KLARA_DIAGT_Q, false, false, true>), grid, blk, 0, st, p, kl);              \
        else if (unitw) hipLaunchKernelGGL((k_diagt<KLARA_SAMPLER_SLICE, NP_, KLARA_DIAGT_Q, false, true, false>), grid, blk, 0, st, p, kl);            \
        else hipLaunchKernelGGL((k_diagt<KLARA_SAMPLER_SLICE, NP_, KLARA_DIAGT_Q, false, false, false>), grid, blk, 0, st, p, kl);                      \
        break;

hipError_t KLARA_DIAGT_FN(klara_launch_diagt_slice)(const KParams* p, const KLaunch& kl, int NP, bool unitw, bool mon, bool tune, dim3 grid, hipStream_t st)
{
    const dim3 blk(256);
    switch (NP) {
        KLARA_DIAGT_NP_MENU_DO(KLARA_DIAGT_SLICE_CASE)
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
