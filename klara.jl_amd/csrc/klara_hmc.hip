// klara_hmc.hip — instantiates the HMC transition kernels (group layout) for gfx950.
#include "klara_launch.h"

hipError_t klara_launch_hmc(const KParams* p, const KLaunch& kl, int mode, int target, int E, int G, dim3 grid, size_t lds,
                            hipStream_t st)
{
    KLARA_DISPATCH_GROUP(k_transitions, KLARA_SAMPLER_HMC);
}
