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
// device, libm calls are not).  `data` is the job's read-only block (klara_desc.custom_data) in device memory.  Two forms of the layout:
//  * D <= 32: one chain per lane (G = 1, E = pow2ceil(D) elements in registers): the user's function sees the lane's registers, no
//    cross-lane reduction exists.  (Also every D <= 256 on request, KLARA_CUSTOM_LANES=1: the vector then lives in scratch beyond 32.)
//  * D > 32, STAGED: G = 4 .. 64 lanes per chain (round 6: 64 — one chain per wavefront — for 513 <= D <= 1024), E = 2 ceil(D / 2G) <= 16 elements per lane in registers.  Proposal normals, sampler
//    arithmetic, running sums and monitors are spread over the chain's lanes like in the built-in group layout; for an evaluation the
//    lanes write their elements into the chain's row of LDS, every lane of the chain calls the user's function on that row (the G
//    evaluations are identical — a wavefront serves 64 / G chains per evaluation instead of 64 — but the vector never sees scratch), the
//    gradient is written to a second row and each lane picks up its own elements.  The sums of the samplers run in the group layout's
//    order (klara_get_layout: kind 0, G, E), the target's value is the user's own.
//
// This file is only ever compiled by the run-time compiler, after klara_kernels.h and the user's source.
#pragma once
#include "klara_custom_compose.h"       // likelihood + prior form (KLARA_USER_LIKELIHOOD_PRIOR): lt = ll + lp, grad = gll + glp

// row length (doubles) of a chain's staging area: NROWS vectors of D rounded up to even (16-byte aligned: the closures' reads of
// neighbouring elements merge into 16-byte LDS reads) plus a pad that keeps the rows of neighbouring chains — the two to four distinct
// addresses of one pass of an LDS instruction — in different banks
#ifdef KLARA_USER_LIKELIHOOD_PRIOR
#define KLARA_CUSTOM_STAGE_ROWS 3          // x, gradloglikelihood, gradlogprior
#else
#define KLARA_CUSTOM_STAGE_ROWS 2          // x, gradlogtarget
#endif
__host__ __device__ inline int klara_custom_stage_stride(int D, int nrows)
{
    const int s = nrows * ((D + 1) & ~1) + 2;
    return (2 * s) % 64 == 0 ? s + 2 : s;
}

template <int E>
struct CustomTarget {
    const double* data; long long ndata; int D;
    double* xs; double* gs; double* ts;      // STAGED: the chain's rows of LDS (value, gradient, second gradient of the likelihood + prior form)
    bool staged;
    mutable double lt_full;                  // STAGED: the user's log-target (every lane of the chain holds it; nothing to reduce)
    static __device__ __forceinline__ size_t lds_bytes(const KParams&) { return 0; }
    __device__ __forceinline__ void init(const KParams& p, const LaneCtx<E>& cx, double* lds)
    {
        data = (const double*)p.cdata; ndata = p.cndata; D = p.D;
        staged = cx.G > 1; lt_full = 0.0;
        const int dp = (p.D + 1) & ~1;
        const int row = (int)(threadIdx.x >> 6) * (64 / cx.G) + cx.lane / cx.G;
        xs = (double*)__builtin_assume_aligned(lds + (size_t)row * klara_custom_stage_stride(p.D, KLARA_CUSTOM_STAGE_ROWS), 16);
        gs = (double*)__builtin_assume_aligned(xs + dp, 16); ts = (double*)__builtin_assume_aligned(gs + dp, 16);
    }
    // the wavefront's LDS accesses execute in program order; the fence keeps the compiler from moving a lane's reads of other lanes'
    // elements across its own writes
    static __device__ __forceinline__ void stage_fence()
    {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    __device__ __forceinline__ void stage(const LaneCtx<E>& cx, const double (&x)[E]) const
    {
KLARA_PRAGMA_UNROLL_E
        for (int e = 0; e < E; ++e) if (cx.i0 + e < D) xs[cx.i0 + e] = x[e];
        stage_fence();
    }
    // :monitor => [:loglikelihood, :logprior] (iterate/MALA.jl:104-109): the two parts at the saved state
    __device__ __forceinline__ void parts(const LaneCtx<E>& cx, const double (&x)[E], double& ll, double& lp) const
    {
#ifdef KLARA_USER_LIKELIHOOD_PRIOR
        if (staged) {
            stage(cx, x);
            ll = klara_user_loglikelihood(xs, D, data, ndata);
            lp = klara_user_logprior(xs, D, data, ndata);
            stage_fence();
        } else {
            ll = klara_user_loglikelihood(x, D, data, ndata);
            lp = klara_user_logprior(x, D, data, ndata);
        }
#else
        ll = 0.0; lp = 0.0;
#endif
    }
    template <bool WANT_LT, bool WANT_GRAD>
    __device__ __forceinline__ void eval(const LaneCtx<E>& cx, const double (&x)[E], double& ltpart, double (&g)[E]) const
    {
        if (staged) {
            stage(cx, x);
            if (WANT_LT) { lt_full = klara_user_logtarget(xs, D, data, ndata); ltpart = 0.0; }
            if (WANT_GRAD) {
#ifdef KLARA_CUSTOM_NOGRAD
KLARA_PRAGMA_UNROLL_E
                for (int e = 0; e < E; ++e) g[e] = 0.0;
#elif defined(KLARA_USER_LIKELIHOOD_PRIOR)
                klara_user_gradloglikelihood(xs, D, data, ndata, gs);
                klara_user_gradlogprior(xs, D, data, ndata, ts);
                stage_fence();
KLARA_PRAGMA_UNROLL_E
                for (int e = 0; e < E; ++e) g[e] = cx.i0 + e < D ? gs[cx.i0 + e] + ts[cx.i0 + e] : 0.0;     // (klara_custom_compose.h: one addition per element)
#else
                klara_user_gradlogtarget(xs, D, data, ndata, gs);
                stage_fence();
KLARA_PRAGMA_UNROLL_E
                for (int e = 0; e < E; ++e) g[e] = cx.i0 + e < D ? gs[cx.i0 + e] : 0.0;
#endif
            }
            stage_fence();
            return;
        }
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
    __device__ __forceinline__ double finalize(double red) const { return staged ? lt_full : red; }
};
template <int E> struct TargetSel<KLARA_TARGET_CUSTOM, E> { using type = CustomTarget<E>; };
