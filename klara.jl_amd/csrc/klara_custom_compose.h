/* klara_custom_compose.h — likelihood + prior form of a user-defined target (KLARA_TARGET_CUSTOM).
 *
 * The reference's parameter takes the target either whole (logtarget / gradlogtarget closures) or as a likelihood and a prior:
 *     logtarget!(state)     = loglikelihood!(state); logprior!(state); state.logtarget = state.loglikelihood + state.logprior
 *     gradlogtarget!(state) = gradloglikelihood!; gradlogprior!;       state.gradlogtarget = gradloglikelihood + gradlogprior
 * (src/variables/parameters/BasicContMuvParameter.jl:174-201: the `isa(args[i-2], Function) && isa(ppfield, Function)` branch,
 * likelihood first, then prior, then the sum).  A user source that starts with
 *     #define KLARA_USER_LIKELIHOOD_PRIOR 1
 * defines, instead of klara_user_logtarget / klara_user_gradlogtarget,
 *     KLARA_USER_FN double klara_user_loglikelihood(const double* x, int D, const double* data, long long ndata);
 *     KLARA_USER_FN double klara_user_logprior(const double* x, int D, const double* data, long long ndata);
 *     KLARA_USER_FN void   klara_user_gradloglikelihood(const double* x, int D, const double* data, long long ndata, double* g);
 *     KLARA_USER_FN void   klara_user_gradlogprior(const double* x, int D, const double* data, long long ndata, double* g);
 * (the two gradients for MALA / HMC only) and this header, compiled right after the user's text by the run-time compiler (device) and
 * by the host C compiler (CPU oracle), supplies the composition.  Plain C, one addition per element: the same bits on both sides. */
#ifndef KLARA_CUSTOM_COMPOSE_H
#define KLARA_CUSTOM_COMPOSE_H
#ifdef KLARA_USER_LIKELIHOOD_PRIOR
KLARA_USER_FN double klara_user_logtarget(const double* x, int D, const double* data, long long ndata)
{
    const double ll = klara_user_loglikelihood(x, D, data, ndata);
    const double lp = klara_user_logprior(x, D, data, ndata);
    return ll + lp;
}
#ifndef KLARA_CUSTOM_NOGRAD
KLARA_USER_FN void klara_user_gradlogtarget(const double* x, int D, const double* data, long long ndata, double* g)
{
    double gl[KLARA_D], gp[KLARA_D];
    klara_user_gradloglikelihood(x, D, data, ndata, gl);
    klara_user_gradlogprior(x, D, data, ndata, gp);
    for (int i = 0; i < KLARA_D; ++i) g[i] = gl[i] + gp[i];
}
#endif
#endif
/* A pair closure (`#define KLARA_USER_PAIR_TARGET 1` + klara_user_pair, include/klara_hip.h) taken as a whole-vector closure — what a job runs on that the
 * pair-transposed kernels do not serve (fewer than 17 dimensions, or the slice sampler): klara_create prefixes the source with
 * `#define KLARA_PAIR_AS_WHOLE 1`, and the sum over the pairs is formed here, pair 0 first, one addition per pair — the same text on the device and in the
 * CPU oracle.  The missing half of an odd D's last pair is passed as 0 and its derivative is dropped (as on the pair kernels). */
#if defined(KLARA_USER_PAIR_TARGET) && defined(KLARA_PAIR_AS_WHOLE)
KLARA_USER_FN double klara_user_logtarget(const double* x, int D, const double* data, long long ndata)
{
    double s = 0.0;
    for (int P = 0; P < (KLARA_D + 1) / 2; ++P) {
        double g0, g1;
        const int full = 2 * P + 1 < KLARA_D;
        s = s + klara_user_pair(x[2 * P], full ? x[2 * P + 1] : 0.0, P, D, data, ndata, &g0, &g1);
    }
    return s;
}
#ifndef KLARA_CUSTOM_NOGRAD
KLARA_USER_FN void klara_user_gradlogtarget(const double* x, int D, const double* data, long long ndata, double* g)
{
    for (int P = 0; P < (KLARA_D + 1) / 2; ++P) {
        double g0 = 0.0, g1 = 0.0;
        const int full = 2 * P + 1 < KLARA_D;
        (void)klara_user_pair(x[2 * P], full ? x[2 * P + 1] : 0.0, P, D, data, ndata, &g0, &g1);
        g[2 * P] = g0;
        if (full) g[2 * P + 1] = g1;
    }
}
#endif
#endif
#endif
