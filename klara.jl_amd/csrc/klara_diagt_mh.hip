// klara_diagt_mh.hip — instantiates the pair-transposed MH kernels (layout kind 3) for gfx950.
#include "klara_launch.h"

hipError_t KLARA_DIAGT_FN(klara_launch_diagt_mh)(const KParams* p, const KLaunch& kl, int NP, bool onestep, bool unitw, bool mon, bool tune, bool da, const KAuto& ka, long long nwaves, hipStream_t st)
{
    KLARA_DISPATCH_DIAGT(KLARA_SAMPLER_MH);
}
