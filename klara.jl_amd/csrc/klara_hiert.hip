// klara_hiert.hip — instantiates the few-lanes-per-chain kernels of the hierarchical target (layout kind 4) for gfx950.
#include "klara_launch.h"
#include "klara_hiert.h"

template <int SAMPLER>
static hipError_t launch_hiert(const KParams* p, const KLaunch& kl, bool mon, bool tune, dim3 grid, hipStream_t st)
{
    const dim3 blk(256);
    if (tune) return klara_go(k_hiert<SAMPLER, 4, 5, true, true>, grid, blk, 0, st, p, kl);
    if (mon) return klara_go(k_hiert<SAMPLER, 4, 5, true, false>, grid, blk, 0, st, p, kl);
    return klara_go(k_hiert<SAMPLER, 4, 5, false, false>, grid, blk, 0, st, p, kl);
}

hipError_t klara_launch_hiert(const KParams* p, const KLaunch& kl, int sampler, int RPL, int NT, bool mon, bool tune, bool da, dim3 grid,
                              hipStream_t st)
{
    if (RPL != 4 || NT < 1) return hipErrorInvalidValue;      // (the kernels read the observation count from KParams; NT = 5 names the instantiation)
    switch (sampler) {
    case KLARA_SAMPLER_MH: return launch_hiert<KLARA_SAMPLER_MH>(p, kl, mon, tune, grid, st);
    case KLARA_SAMPLER_MALA: return launch_hiert<KLARA_SAMPLER_MALA>(p, kl, mon, tune, grid, st);
    case KLARA_SAMPLER_HMC:
        if (da) {
            return klara_go(k_hiert<KLARA_SAMPLER_HMC, 4, 5, true, true, true>, grid, dim3(256), 0, st, p, kl);
        }
        return launch_hiert<KLARA_SAMPLER_HMC>(p, kl, mon, tune, grid, st);
    default: return hipErrorInvalidValue;
    }
}

hipError_t klara_launch_hiert_init(const KParams& p, int RPL, int NT, int needgrad, dim3 grid, hipStream_t st)
{
    if (RPL != 4 || NT < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL((k_hiert_init<4, 5>), grid, dim3(256), 0, st, p, needgrad);
    return hipGetLastError();
}
