// klara_diagt_init.hip — initialize! for layout kind 3 (lt and gradient at X).
#include "klara_launch.h"

hipError_t KLARA_DIAGT_FN(klara_launch_diagt_init)(const KParams& p, int NP, int needgrad, dim3 grid, hipStream_t st)
{
    const dim3 blk(256);
    switch (NP) {
#define X(NP_) case NP_: hipLaunchKernelGGL((k_diagt_init<NP_, KLARA_DIAGT_Q>), grid, blk, 0, st, p, needgrad); break;
        KLARA_DIAGT_NP_MENU_DO(X)
#undef X
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
