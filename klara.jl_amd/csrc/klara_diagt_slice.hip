// klara_diagt_slice.hip — instantiates the pair-transposed slice-sampler kernels (layout kind 3) for gfx950.
#include "klara_launch.h"
#include "klara_diagt_slice.h"
#include <cstdlib>
#ifndef KLARA_SLICEF_DEFAULT_NM
#define KLARA_SLICEF_DEFAULT_NM 1     // (same box, 65,536 x 100: 9.7e10 coordinate updates/s with one machine per lane, 9.5e10 with two: profiles/r5_ab_slice.txt)
#endif

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

// untuned jobs without a history monitor: every lane takes its element pairs through the whole launch on its own (klara_diagt_slice.h)
hipError_t KLARA_DIAGT_FN(klara_launch_diagt_slice_free)(const KParams* p, const KLaunch& kl, int NP, bool unitw, bool mon, const KAuto& ka, long long nwaves, hipStream_t st)
{
    const bool sums = mon;      // (template flag SUMS: a saved-sample monitor — running sums and / or value history — is on)
    if (NP < 1 || NP > KLARA_SLICEF_MAXNP) return hipErrorInvalidValue;
    const dim3 grid((unsigned)((nwaves + 3) / 4)), blk(256);
    static const int nm = getenv("KLARA_SLICE_MACHINES") ? atoi(getenv("KLARA_SLICE_MACHINES")) : KLARA_SLICEF_DEFAULT_NM;
    // dynamic LDS: the widths of the job's 2 NP Q element slots, and weights + means for a non-unit diagonal
    const size_t lds = (size_t)(unitw ? 1 : 3) * 2 * NP * KLARA_DIAGT_Q * sizeof(double);
#define KLARA_SLICEF_GO(U, S)                                                                                          \
    (nm == 1 ? klara_go(k_diagt_slice_free<KLARA_DIAGT_Q, U, S, 1>, grid, blk, lds, st, p, kl, ka, NP)                  \
             : klara_go(k_diagt_slice_free<KLARA_DIAGT_Q, U, S, 2>, grid, blk, lds, st, p, kl, ka, NP))
    const hipError_t e = unitw ? (sums ? KLARA_SLICEF_GO(true, true) : KLARA_SLICEF_GO(true, false)) : (sums ? KLARA_SLICEF_GO(false, true) : KLARA_SLICEF_GO(false, false));
#undef KLARA_SLICEF_GO
    if (e != hipSuccess || klara_attr_query != nullptr) return e;
    // the new state's log-target in the layout's order (the kernel above deals the elements to the lanes round robin): one wavefront per chain group
    return unitw ? klara_go(k_diagt_hist_lt<KLARA_DIAGT_Q, true, true>, grid, blk, 0, st, p, kl, NP, 0LL, 1)
                 : klara_go(k_diagt_hist_lt<KLARA_DIAGT_Q, false, true>, grid, blk, 0, st, p, kl, NP, 0LL, 1);
}

// log-target history of the `ncols` states a launch of the kernel above saved (columns col0 ...), from their saved values
hipError_t KLARA_DIAGT_FN(klara_launch_diagt_hist_lt)(const KParams* p, const KLaunch& kl, int NP, bool unitw, long long col0, int ncols, long long ngroups, hipStream_t st)
{
    if (ncols <= 0 || ngroups <= 0) return hipSuccess;
    const long long nwaves = ngroups * ncols;
    const dim3 grid((unsigned)((nwaves + 3) / 4)), blk(256);
    return unitw ? klara_go(k_diagt_hist_lt<KLARA_DIAGT_Q, true>, grid, blk, 0, st, p, kl, NP, col0, ncols)
                 : klara_go(k_diagt_hist_lt<KLARA_DIAGT_Q, false>, grid, blk, 0, st, p, kl, NP, col0, ncols);
}
