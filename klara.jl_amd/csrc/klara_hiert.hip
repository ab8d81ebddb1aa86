// klara_hiert.hip — instantiates the few-lanes-per-chain HMC kernels of the hierarchical target (layout kind 4) for gfx950.
#include "klara_launch.h"
#include "klara_hiert.h"

hipError_t klara_launch_hiert_hmc(const KParams* p, const KLaunch& kl, int RPL, int NT, bool mon, bool tune, bool da, dim3 grid, hipStream_t st)
{
    const dim3 blk(256);
    if (RPL != 4 || NT != 5) return hipErrorInvalidValue;
    if (da) hipLaunchKernelGGL((k_hiert_hmc<4, 5, true, true, true>), grid, blk, 0, st, p, kl);
    else if (tune) hipLaunchKernelGGL((k_hiert_hmc<4, 5, true, true>), grid, blk, 0, st, p, kl);
    else if (mon) hipLaunchKernelGGL((k_hiert_hmc<4, 5, true, false>), grid, blk, 0, st, p, kl);
    else hipLaunchKernelGGL((k_hiert_hmc<4, 5, false, false>), grid, blk, 0, st, p, kl);
    return hipGetLastError();
}

hipError_t klara_launch_hiert_init(const KParams& p, int RPL, int NT, int needgrad, dim3 grid, hipStream_t st)
{
    if (RPL != 4 || NT != 5) return hipErrorInvalidValue;
    hipLaunchKernelGGL((k_hiert_init<4, 5>), grid, dim3(256), 0, st, p, needgrad);
    return hipGetLastError();
}
