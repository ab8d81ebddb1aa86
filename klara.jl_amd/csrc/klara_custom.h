// klara_custom.h — user-defined target for the group-layout kernels, compiled at run time (hiprtc) together with
// klara_kernels.h and the user's source.  This is the device form of the reference's target closures
// (`BasicContMuvParameter(:p, logtarget=f, gradlogtarget=g)`, src/variables/parameters/BasicContMuvParameter.jl:174-201,
// 264-279; `uptogradlogtarget!` falls back to `logtarget!; gradlogtarget!`, :270-274): the user's source text defines
//
//   KLARA_USER_FN double klara_user_logtarget(const double* x, int D, const double* data, long long ndata);
//   KLARA_USER_FN void   klara_user_gradlogtarget(const double* x, int D, const double* data, long long ndata, double* g);
//
// in the C subset both hipcc and a host C compiler accept (KLARA_D is predefined to the job's dimension so that loops
// unroll and x / g stay in registers; kd_exp, kd_log, kd_fma, kd_erf and IEEE + - * / sqrt are bit-reproducible on host and
// device, libm calls are not).  One chain per lane (G = 1, E = pow2ceil(D) <= 256 elements: in registers up to 32, in scratch beyond; above 128 elements
// the element loops are not unrolled, klara_jit.hip), so the user's
// function sees the whole parameter vector and no cross-lane reduction exists; `data` is the job's read-only block
// (klara_desc.custom_data) in device memory.
//
// This file is only ever compiled by the run-time compiler, after klara_kernels.h and the user's source.
#pragma once
#include "klara_custom_compose.h"       // likelihood + prior form (KLARA_USER_LIKELIHOOD_PRIOR): lt = ll + lp, grad = gll + glp

template <int E>
struct CustomTarget {
    // :monitor => [:loglikelihood, :logprior] (iterate/MALA.jl:104-109): the two parts at the saved state
    __device__ __forceinline__ void parts(const double (&x)[E], double& ll, double& lp) const
    {
#ifdef KLARA_USER_LIKELIHOOD_PRIOR
        ll = klara_user_loglikelihood(x, D, data, ndata);
        lp = klara_user_logprior(x, D, data, ndata);
#else
        ll = 0.0; lp = 0.0;
#endif
    }
    const double* data; long long ndata; int D;
    static __device__ __forceinline__ size_t lds_bytes(const KParams&) { return 0; }
    __device__ __forceinline__ void init(const KParams& p, const LaneCtx<E>&, double*)
    {
        data = (const double*)p.cdata; ndata = p.cndata; D = p.D;
    }
    template <bool WANT_LT, bool WANT_GRAD>
    __device__ __forceinline__ void eval(const LaneCtx<E>&, const double (&x)[E], double& ltpart, double (&g)[E]) const
    {
        if (WANT_LT) ltpart = klara_user_logtarget(x, D, data, ndata);
        if (WANT_GRAD) {
#ifdef KLARA_CUSTOM_NOGRAD                                           // MH / slice sampler: no gradient closure is required
KLARA_PRAGMA_UNROLL_E
            for (int e = 0; e < E; ++e) g[e] = 0.0;
#else
            klara_user_gradlogtarget(x, D, data, ndata, g);
KLARA_PRAGMA_UNROLL_E
            for (int e = 0; e < E; ++e) if (e >= D) g[e] = 0.0;     // padding elements stay exactly zero
#endif
        }
    }
    __device__ __forceinline__ double finalize(double red) const { return red; }
};
template <int E> struct TargetSel<KLARA_TARGET_CUSTOM, E> { using type = CustomTarget<E>; };
