// klara_diagt_slice.hip — instantiates the pair-transposed slice-sampler kernels (layout kind 3) for gfx950.
#include "klara_launch.h"

#define KLARA_DIAGT_SLICE_CASE(NP_)                                                                                       \
    case NP_:                                                                                                              \
        if (tune && unitw) e_ = diagt_go<KLARA_SAMPLER_SLICE, NP_, KLARA_DIAGT_Q, false, true, true, true>(p, kl, ka, nwaves, st);        \
        else if (tune) e_ = diagt_go<KLARA_SAMPLER_SLICE, NP_, KLARA_DIAGT_Q, false, false, true, true>(p, kl, ka, nwaves, st);       \
        else if (mon && unitw) e_ = diagt_go<KLARA_SAMPLER_SLICE, NP_, KLARA_DIAGT_Q, false, true, true>(p, kl, ka, nwaves, st);      \
        else if (mon) e_ = diagt_go<KLARA_SAMPLER_SLICE, NP_, KLARA_DIAGT_Q, false, false, true>(p, kl, ka, nwaves, st);              \
        else if (unitw) e_ = diagt_go<KLARA_SAMPLER_SLICE, NP_, KLARA_DIAGT_Q, false, true, false>(p, kl, ka, nwaves, st);            \
        else e_ = diagt_go<KLARA_SAMPLER_SLICE, NP_, KLARA_DIAGT_Q, false, false, false>(p, kl, ka, nwaves, st);                      \
        break;

hipError_t KLARA_DIAGT_FN(klara_launch_diagt_slice)(const KParams* p, const KLaunch& kl, int NP, bool unitw, bool mon, bool tune, const KAuto& ka, long long nwaves, hipStream_t st)
{
    hipError_t e_ = hipSuccess;
    switch (NP) {
        KLARA_DIAGT_NP_MENU_DO(KLARA_DIAGT_SLICE_CASE)
    default: return hipErrorInvalidValue;
    }
    return e_;
}
