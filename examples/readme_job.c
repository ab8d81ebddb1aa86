/* readme_job.c — the reference's README job (README.md:23-66: MH on lt = -dot(z, z), 10,000 steps, burn-in 1,000) for N chains
 * through the C ABI of libklara_hip.so, in plain C99.  Shows the call sequence a binding follows:
 *   klara_create -> klara_set_state -> klara_run -> klara_get_chain_sums / klara_get_accept_counts / klara_gather_moments -> klara_destroy.
 *
 *   gcc -std=c99 -O2 -Iinclude examples/readme_job.c -Lklara.jl_amd/lib -lklara_hip -Wl,-rpath,$PWD/klara.jl_amd/lib -o readme_job
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "klara_hip.h"

#define CHECK(call)                                                                        \
    do {                                                                                   \
        klara_status st_ = (call);                                                         \
        if (st_ != KLARA_OK) {                                                             \
            fprintf(stderr, "%s: %s\n", #call, klara_strerror(st_));                       \
            return 1;                                                                      \
        }                                                                                  \
    } while (0)

int main(int argc, char** argv)
{
    const long long nchains = argc > 1 ? atoll(argv[1]) : 4096;
    const int D = 2;
    const double sigma[2] = { 1.0, 1.0 };                 /* MH(ones(2)) */
    klara_desc d;
    klara_handle* job = NULL;
    double *x0, *sum, *sumsq;
    uint64_t *naccept, nsteps_done = 0;
    int64_t nsaved = 0;
    long long c;
    double mean0 = 0.0, mean1 = 0.0, var0 = 0.0, acc = 0.0;
    double pmean[2], pm2[2];
    uint64_t psamples = 0, paccept = 0, ptrans = 0, pchains = 0;

    memset(&d, 0, sizeof d);
    d.struct_size = (uint32_t)sizeof d;
    d.abi_version = KLARA_ABI_VERSION;
    d.sampler = KLARA_SAMPLER_MH;
    d.target = KLARA_TARGET_GAUSS_DIAG;                   /* w = 1, mu = 0, c = 0: lt = -dot(z, z) */
    d.tuner = KLARA_TUNER_VANILLA;
    d.tuner_mode = KLARA_TUNE_PER_CHAIN;
    d.nchains = nchains;
    d.ndims = D;
    d.mh_sigma = sigma;
    d.period = 100;
    d.nsteps = 10000;                                     /* BasicMCRange(nsteps=10000, burnin=1000) */
    d.burnin = 1000;
    d.thinning = 1;
    d.seed = 20260927u;
    d.monitor = KLARA_MON_SUMMARIES;                      /* per-chain running sums of the saved samples */
    CHECK(klara_create(&d, &job));

    x0 = (double*)malloc(sizeof(double) * (size_t)nchains * D);
    sum = (double*)malloc(sizeof(double) * (size_t)nchains * D);
    sumsq = (double*)malloc(sizeof(double) * (size_t)nchains * D);
    naccept = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)nchains);
    if (!x0 || !sum || !sumsq || !naccept) return 1;
    for (c = 0; c < nchains; ++c) { x0[2 * c] = 5.1; x0[2 * c + 1] = -0.9; }   /* v0 = Dict(:p => [5.1, -0.9]) */
    CHECK(klara_set_state(job, x0));
    CHECK(klara_run(job, d.nsteps));
    CHECK(klara_get_chain_sums(job, sum, sumsq, &nsaved));
    CHECK(klara_get_accept_counts(job, naccept, &nsteps_done));
    for (c = 0; c < nchains; ++c) {
        mean0 += sum[2 * c] / (double)nsaved;
        mean1 += sum[2 * c + 1] / (double)nsaved;
        var0 += sumsq[2 * c] / (double)nsaved;
        acc += (double)naccept[c] / (double)nsteps_done;
    }
    printf("%lld chains x %lld saved samples: mean = (%.4f, %.4f) (truth 0), E[z1^2] = %.4f (truth 0.5), acceptance = %.3f\n",
           nchains, (long long)nsaved, mean0 / (double)nchains, mean1 / (double)nchains, var0 / (double)nchains, acc / (double)nchains);
    /* the pooled posterior moments formed on the device (per-chain (n, mean, M2), Chan's merge): mean(chain) / var over all chains at once;
     * with a klara_comm instead of NULL the same call merges the chains of every GPU of the job */
    CHECK(klara_gather_moments(job, NULL, pmean, pm2, &psamples, &paccept, &ptrans, &pchains));
    printf("pooled over %llu samples: mean = (%.5f, %.5f), var = (%.5f, %.5f) (truth 0.5), acceptance = %.4f\n", (unsigned long long)psamples,
           pmean[0], pmean[1], pm2[0] / (double)psamples, pm2[1] / (double)psamples, (double)paccept / (double)ptrans);
    CHECK(klara_destroy(job));
    free(x0); free(sum); free(sumsq); free(naccept);
    return 0;
}
