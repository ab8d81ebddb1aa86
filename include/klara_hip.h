/* klara_hip.h — C ABI of libklara_hip.so: the many-chain MCMC transition path of Klara.jl on MI355X.
 *
 * What this replaces.  Klara.jl has no FFI (it is 100 % Julia); the hot path sits behind Julia
 * multiple dispatch:
 *     run(job::BasicMCJob)                               src/jobs/BasicMCJob.jl:212-244
 *       iterate!(job, typeof(job.sampler), variate_form)  src/jobs/BasicMCJob.jl:224
 *         iterate!(::BasicMCJob, ::Type{MH|MALA|HMC|SliceSampler}, ::Type{Multivariate})
 *                                                         src/samplers/iterate/{MH,MALA,HMC,SliceSampler}.jl
 *         job.parameter.logtarget! / gradlogtarget! / uptogradlogtarget!
 *                                                         src/variables/parameters/BasicContMuvParameter.jl:174-279
 * One klara_handle stands for N independent BasicMCJobs (N chains of one model); klara_run() is the
 * `for i in 1:nsteps iterate!(...)` loop of BasicMCJob.jl:219-238 executed for all chains at once on
 * the GPU.  The Julia-side binding a maintainer would add is shown in INTEGRATION.md (ccall stubs).
 *
 * Conventions: plain C, caller-owned host buffers copied at the call, library-owned device buffers
 * inside the handle, every entry point returns klara_status (never aborts).  Matrices are row-major
 * (nchains x ndims): chain c's D-vector is contiguous — the transpose of Klara's per-chain
 * NState.value (ndims x nsaved, src/nstates/ParameterNStates/BasicContMuvParameterNState.jl:1-21);
 * klara_get_chain() returns one chain in Klara's own column-major layout.
 *
 * There is no CPU implementation behind this ABI: if no HIP device is present klara_create() fails
 * with KLARA_ERR_HIP.
 */
#ifndef KLARA_HIP_H
#define KLARA_HIP_H

#ifndef __HIPCC_RTC__
#include <stdint.h>
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define KLARA_ABI_VERSION 6   /* 6: Box-Muller angles at the centres of their 2^20 cells (version 5 used the left edges, which put mass 2^-19 of every
                                 normal on the coordinate axes), klara_selftest_transition_normals; same structs as 4 and 5, but a given seed draws
                                 different chains than a version-5 library.  5: one Philox block makes four normals, 44-bit accept uniform.
                                 4: klara_gather_moments, klara_desc.sparse_moves 0 = device-decided */
/* THE RANDOM STREAM IS FROZEN AT VERSION 6.  What a (seed, chain, transition) draws — the counter layout (detmath.h kd_stream_block), the
 * pair -> block-half map (kd_pair_block / kd_pair_half), the 44-bit radius / 20-bit centred angle Box-Muller (kd_normal_pair_w), the accept
 * uniform's slot ceil(D/2) and the slice sampler's slots (kd_slice_attempt_slot) — is pinned by known-answer values committed under
 * tests/golden/stream_kat.json (host build, device build and the independent NumPy restatement must all reproduce them), and the HIP
 * path is tied to the LITERAL Julia arithmetic and to the NumPy mirror on every accept decision at the BASELINE sizes
 * (tests/test_gpu_literal.py), with the joint law of a transition's normals tested in tests/test_stream_joint.py.  A change to the draw
 * schedule, to detmath.h's transforms or to the oracle's arithmetic needs those three green BEFORE and AFTER, and a new version number. */
/* klara_desc.steps_per_launch = 0 selects this many transitions per kernel launch (launches also end at the pooled tuner's
 * events and at batch boundaries of the streaming batch means, whichever comes first) */
#define KLARA_DEFAULT_STEPS_PER_LAUNCH 32
/* ... and for the slice-sampler jobs the free-running kernel serves (klara.jl_amd/csrc/klara_diagt_slice.h: the lanes run out of lockstep, a wavefront
 * waits for its slowest lane once per element slot and launch): diagonal Gaussian, 17 <= D <= 1024, untuned (VanillaMCTuner per chain, not verbose),
 * monitors among the accept diagnostics, the running sums, the value history (ring or not), the log-target history (with the values kept) and
 * the streaming autocovariances.  Ring planning and the cadence of the accept rows follow this length.  Every other slice job: 32. */
#define KLARA_DEFAULT_STEPS_PER_LAUNCH_SLICE 128
#define KLARA_LOGIT_MAX_LDS_DOUBLES 18432u   /* 144 KB of the 160 KB of LDS of a compute unit */

typedef enum klara_status {
    KLARA_OK = 0,
    KLARA_ERR_INVALID_ARG = 1,     /* mirrors the reference's @assert argument checks                */
    KLARA_ERR_NONFINITE_INIT = 2,  /* MH.jl:83 / MALA.jl:83-89 / HMC.jl:113-119: non-finite start     */
    KLARA_ERR_HIP = 3,             /* HIP runtime failure or no device                                */
    KLARA_ERR_NOMEM = 4,
    KLARA_ERR_UNSUPPORTED = 5,     /* valid Klara option that this build does not cover (see DESIGN)  */
    KLARA_ERR_STATE = 6,           /* call order (e.g. run before set_state)                          */
    KLARA_ERR_SLICE_STUCK = 7,     /* iterate/SliceSampler.jl:102 "Shrunk to current position ..." (or 16,383 step-out / shrink attempts): a chain that
                                      raised it stops at the state BEFORE the coordinate update that failed (its X and log-target belong together; the
                                      reference throws out of run(job) at that point), the job's other chains go on; reset the job */
    KLARA_ERR_COMPILE = 8          /* CUSTOM target: the user's source did not compile (klara_compile_log) */
} klara_status;

/* src/samplers/{MH,MALA,HMC,SliceSampler}.jl */
typedef enum klara_sampler {
    KLARA_SAMPLER_MH = 0,      /* MH(sigma): symmetric normal random walk, MH.jl:63-66            */
    KLARA_SAMPLER_MALA = 1,    /* MALA(driftstep), MALA.jl:61-70                                   */
    KLARA_SAMPLER_HMC = 2,     /* HMC(leapstep, nleaps), HMC.jl:89-100                             */
    KLARA_SAMPLER_SLICE = 3    /* SliceSampler(widths, stepout), SliceSampler.jl:22-34             */
} klara_sampler;

/* Target families evaluated on device (stand-ins for the user closures of
 * BasicContMuvParameter.jl:383-411; arbitrary Julia closures cannot run on the GPU). */
typedef enum klara_target {
    /* lt = c - sum_i w_i (x_i - mu_i)^2 ; grad_i = -2 w_i (x_i - mu_i).
     * README.md:23,155 is w=1, mu=0, c=0; MvNormal(mu, sigma I) of test/BasicContMuvParameter.jl:39-56
     * is w = 1/(2 sigma^2), c = -D/2 log(2 pi) - D log sigma. */
    KLARA_TARGET_GAUSS_DIAG = 0,
    /* lt = c - 1/2 (x-mu)' P (x-mu) ; grad = -P (x-mu).  P = dense precision matrix (D x D).  D <= 128: FP64 matrix cores (all four
     * samplers; P in LDS); D = 129..256: all four stay on the matrix cores (P streamed from memory; HMC's momentum in LDS: 64 TFLOP/s
     * at D = 256); D = 257..1024 (round 6; refused before): all four samplers with the tile of 16 chains on a workgroup of 4 / 8 wavefronts that
     * deal the row tiles of P evenly (layout kind 6, klara_dense_split.h: HMC L = 10 at 56 .. 71 TFLOP/s); D > 1024 is KLARA_ERR_UNSUPPORTED. */
    KLARA_TARGET_GAUSS_DENSE = 1,
    /* Bayesian logistic regression of doc/examples/swiss/MALA/analytical.jl:11-18:
     * lt = dot(Xp, y) - sum(log(1+exp(Xp))) - 0.5 (p.p/lambda + D log(2 pi lambda)).  Up to 16 parameters: the data rows in LDS, dealt to 4 lanes
     * per chain; 17 .. 256 parameters (and 9 .. 16 with more rows than the LDS holds), every sampler: X p and X' (y - 1/(1+exp(-Xp))) of 16 chains
     * per wavefront on the FP64 matrix cores, X streamed from memory — any number of rows (round 6; layout kind 5).  (KLARA_LOGIT_NO_MFMA=1 in the
     * environment: the same closures through the run-time compiled path, as in rounds 1-5.)  257 .. 1024 parameters: that closure form, one chain
     * per lane with the vector in scratch (correct, slow; round 6 — refused before). */
    KLARA_TARGET_LOGISTIC = 2,
    /* Hierarchical normal growth-curve model for data/rats/{weight,age}.csv (BASELINE cfg 5).  The reference
     * ships the data but no model (doc/examples/rats/Gibbs.jl:1-7 is a stub), so the target is builder-defined:
     * the BUGS "Rats" model  Y_ij ~ N(alpha_i + beta_i xc_j, sigma_c^2),  alpha_i ~ N(alpha_c, sigma_a^2),
     * beta_i ~ N(beta_c, sigma_b^2),  alpha_c, beta_c ~ N(0, 1/prior_prec),  1/sigma_k^2 ~ Gamma(a, b),
     * sampled in theta = (alpha_1, beta_1, ..., alpha_R, beta_R, alpha_c, beta_c, log sigma_c, log sigma_a,
     * log sigma_b), D = 2R + 5 (65 for the 30 rats).  See DESIGN.md for the log-density. */
    KLARA_TARGET_HIER_NORMAL = 3,
    /* User-defined target — the device form of BasicContMuvParameter(:p, logtarget=f, gradlogtarget=g)
     * (BasicContMuvParameter.jl:174-201,264-279; uptogradlogtarget! = logtarget!; gradlogtarget!, :270-274).
     * klara_desc.custom_src is source text in the C subset accepted by hipcc and by a host C compiler; it defines
     *   KLARA_USER_FN double klara_user_logtarget(const double* x, int D, const double* data, long long ndata);
     *   KLARA_USER_FN void   klara_user_gradlogtarget(const double* x, int D, const double* data, long long ndata,
     *                                                 double* g);           (needed by MALA / HMC only)
     * and is compiled for gfx950 at klara_create (hiprtc) into the group-layout transition kernels (D <= 1024 — round 6: up to 64 lanes x 16 elements of the staged form; 256 before —; KLARA_D is predefined
     * to D so that loops unroll): up to D = 32 one chain per lane, the whole vector in the lane's registers; beyond, 4 .. 32 lanes per
     * chain — normals, sampler arithmetic and sums spread over them — and for an evaluation the closure reads the vector from the chain's
     * row of LDS, every lane of the chain evaluating it identically (KLARA_CUSTOM_LANES=1 in the environment keeps one chain per lane
     * at any D: the vector then lives in scratch, which only a closure far costlier than the sampler's own work repays).  kd_exp, kd_log,
     * kd_fma, kd_erf (klara.jl_amd/csrc/detmath.h) and IEEE + - * / sqrt give the same bits on host and device;
     * `data` is custom_data (custom_ndata doubles, copied to the device at create).
     * Likelihood + prior form (BasicContMuvParameter(:p, loglikelihood=..., logprior=..., gradloglikelihood=..., gradlogprior=...),
     * BasicContMuvParameter.jl:174-201): a source that starts with `#define KLARA_USER_LIKELIHOOD_PRIOR 1` defines
     * klara_user_loglikelihood, klara_user_logprior (and klara_user_gradloglikelihood, klara_user_gradlogprior for MALA / HMC)
     * instead; the library composes logtarget = loglikelihood + logprior and the gradients' elementwise sum
     * (klara.jl_amd/csrc/klara_custom_compose.h) and can keep both parts per saved step (KLARA_MON_HIST_LLLP).
     * Pair form, for targets that are a sum of terms of one or two neighbouring coordinates (every separable target, pairwise-coupled
     * ones, the README closure): a source that starts with `#define KLARA_USER_PAIR_TARGET 1` defines instead
     *   KLARA_USER_FN double klara_user_pair(double x0, double x1, int pair, int D, const double* data, long long ndata,
     *                                        double* g0, double* g1);
     * with logtarget(x) = sum over pairs P of klara_user_pair(x[2P], x[2P+1], P, ...) and (*g0, *g1) the pair's two partial derivatives
     * (for the half pair of an odd D, x1 is 0 and *g1 is ignored; the function is called for the real pairs only, 0 <= pair < ceil(D/2), so it may
     * index `data` by pair or by coordinate).  Such a job runs on the few-lanes-per-chain kernels of the diagonal
     * Gaussian (layout kind 3: 8 / 16 / 32 / 64 lanes per chain, 17 <= D <= 1024; MH, MALA, HMC with every tuner, the running sums and the
     * value / logtarget / gradlogtarget histories; klara_get_layout reports the summation order: a lane adds its pairs' terms in
     * ascending order, then the butterfly over the chain's lanes) instead of holding the whole vector in one lane: at D = 100 the
     * README closure runs at 3.5e9 transitions/s in this form and at 2.1e8 in the whole-vector form.  A pair-form job those kernels do
     * not serve — D < 17, or the slice sampler — is taken as a whole-vector closure whose logtarget is the sum of the pairs' terms,
     * pair 0 first (klara_custom_compose.h; D <= 1024). */
    KLARA_TARGET_CUSTOM = 4
} klara_target;

/* src/tuners/{VanillaMCTuner,AcceptanceRateMCTuner}.jl */
typedef enum klara_tuner {
    KLARA_TUNER_VANILLA = 0,
    KLARA_TUNER_ACCEPT_RATE = 1,
    /* DualAveragingMCTuner (src/tuners/DualAveragingMCTuner.jl:54-101), HMC only (HMC.jl:124-133): Nesterov dual
     * averaging of the leapfrog step during the first da_nadapt transitions, nleaps = max(1, round(lambda/step))
     * per transition (iterate/HMC.jl:142-144), step = eps_bar afterwards (iterate/HMC.jl:246-248).  Per chain. */
    KLARA_TUNER_DUAL_AVERAGING = 2
} klara_tuner;

typedef enum klara_tuner_mode {
    KLARA_TUNE_PER_CHAIN = 0,   /* reference semantics: one BasicMCTune per job = per chain      */
    KLARA_TUNE_POOLED = 1       /* one tune state per GPU, rate pooled over the GPU's chains      */
} klara_tuner_mode;

/* monitor flags (jobs.jl:9-43 outopts :monitor / :diagnostics / :destination) */
#define KLARA_MON_ACCEPT    0x1u  /* keep the per-step accept diagnostics (u8 per step per chain)  */
#define KLARA_MON_HISTORY   0x2u  /* :destination=>:nstate — keep value at every postrange step    */
#define KLARA_MON_SUMMARIES 0x4u  /* accumulate sum x, sum x^2 over postrange steps on device      */
#define KLARA_MON_HIST_LT   0x8u  /* :monitor=>[:logtarget]: keep logtarget at every postrange step */
#define KLARA_MON_HIST_GRAD 0x10u /* :monitor=>[:gradlogtarget] (MALA/HMC only)                    */
#define KLARA_MON_HIST_LLLP 0x20u /* :monitor=>[:loglikelihood, :logprior]: the two parts of a likelihood + prior user target
                                     (KLARA_TARGET_CUSTOM whose source defines KLARA_USER_LIKELIHOOD_PRIOR) at every postrange step */

typedef struct klara_desc {
    uint32_t struct_size;        /* = sizeof(klara_desc), ABI check                                  */
    uint32_t abi_version;        /* = KLARA_ABI_VERSION                                              */

    int32_t  sampler;            /* klara_sampler                                                    */
    int32_t  target;             /* klara_target                                                     */
    int32_t  tuner;              /* klara_tuner                                                      */
    int32_t  tuner_mode;         /* klara_tuner_mode                                                 */

    int64_t  nchains;            /* chains owned by this handle (this GPU's shard)                   */
    int64_t  chain_offset;       /* global id of local chain 0 (RNG subsequence = offset + local id) */
    int32_t  ndims;              /* D                                                                */
    int32_t  device;             /* HIP device ordinal                                               */

    /* sampler parameters */
    const double* mh_sigma;      /* MH: proposal std-devs, D doubles (MH.jl:63: MvNormal(x, sigma))   */
    double   driftstep;          /* MALA (MALA.jl:65: > 0)                                           */
    double   leapstep;           /* HMC  (HMC.jl:94: > 0)                                            */
    int32_t  nleaps;             /* HMC  (HMC.jl:95: > 0)                                            */
    int32_t  slice_stepout;      /* SliceSampler.stepout                                             */
    const double* slice_widths;  /* SliceSampler.widths, D doubles (> 0, SliceSampler.jl:27)         */

    /* tuner (AcceptanceRateMCTuner.jl:38-44; VanillaMCTuner: only period/verbose are used) */
    double   targetrate;         /* in (0,1)                                                         */
    double   score_k;            /* steepness of logistic_rate_score (default 7)                     */
    int32_t  period;             /* > 0, default 100                                                 */
    int32_t  verbose;            /* counts proposals the way the reference's verbose tuners do      */
    /* DualAveragingMCTuner(targetrate, nadapt; e0bar=1, h0bar=0, gamma=0.05, t0=10, kappa=0.75) */
    int64_t  da_nadapt;          /* > 0                                                              */
    double   da_eps0bar;         /* > 0                                                              */
    double   da_h0bar;
    double   da_gamma;
    double   da_kappa;
    int32_t  da_t0;              /* > 0                                                              */
    int32_t  tuner_score;        /* AcceptanceRate score: 0 logistic_rate_score(x, score_k) (default k 7), 1 erf_rate_score(x, score_k) (default k 3) */

    /* range (BasicMCRange.jl:17-36) */
    int64_t  nsteps;
    int64_t  burnin;
    int64_t  thinning;

    /* target parameters (host pointers, copied at create) */
    const double* gauss_w;       /* DIAG: D weights (NULL = all 1)                                   */
    const double* gauss_mu;      /* DIAG/DENSE: D means (NULL = 0)                                   */
    double   gauss_const;        /* DIAG/DENSE: additive constant c                                  */
    const double* gauss_prec;    /* DENSE: D*D row-major precision matrix                            */
    const double* logit_X;       /* LOGISTIC: ndata x D row-major design matrix                      */
    const double* logit_y;       /* LOGISTIC: ndata outcomes                                         */
    int32_t  logit_ndata;        /* LOGISTIC: rows.  D <= 8: ndata * (E + 1) <= KLARA_LOGIT_MAX_LDS_DOUBLES, E = 2, 4 or 8 >= D (rows live in LDS);
                                    D = 9..16: the same kernels (E = 16) while the rows fit, the closure form beyond; D = 17..256: any number of
                                    rows (run-time compiled closure form, rows read from memory) */
    int32_t  nstreams;           /* internal HIP streams for independent chain partitions (pair-transposed layout only):
                                    0 = automatic (2 when a partition still fills the GPU), 1..4 = forced        */
    double   logit_lambda;       /* LOGISTIC: prior variance (v[1] of the example)                   */
    const double* hier_Y;        /* HIER_NORMAL: R x T row-major observations                        */
    const double* hier_xc;       /* HIER_NORMAL: T centred covariate values (age_j - 22 for rats)    */
    int32_t  hier_nunits;        /* R (D must equal 2R + 5)                                          */
    int32_t  hier_ntimes;        /* T (<= 16)                                                        */
    double   hier_prior_prec;    /* precision of the N(0, .) priors on alpha_c, beta_c               */
    double   hier_gamma_a;       /* Gamma(a, b) prior on the three precisions                        */
    double   hier_gamma_b;
    const char*   custom_src;    /* CUSTOM: NUL-terminated source text (see KLARA_TARGET_CUSTOM)     */
    const double* custom_data;   /* CUSTOM: read-only data block handed to the user functions, or NULL */
    int64_t  custom_ndata;       /* CUSTOM: doubles in custom_data                                   */
    int64_t  bm_batchlen;        /* > 0: streaming batch means over the saved samples (needs KLARA_MON_SUMMARIES):
                                    every bm_batchlen saved samples a batch of every (chain, dimension) series is closed on
                                    device; klara_get_chain_bm then gives mcvar(:bm) (mcvar.jl:35-41) with no stored history */

    int64_t  hist_ring_cols;     /* > 0: the history monitors (value / logtarget / gradlogtarget / loglikelihood, logprior) keep only
                                    the LAST hist_ring_cols saved steps (a ring): bounded memory for jobs that drain the samples as
                                    they go (the :iostream destination with :flush, jobs.jl:17-29).  0: every saved step is kept. */
    int32_t  acov_maxlag;        /* > 0: lagged cross-products of every (chain, dimension) series are accumulated while sampling
                                    (lags 0..acov_maxlag, at most 127: 3 (maxlag + 1) doubles per series), so that Geyer's initial monotone / positive sequence estimators
                                    (mcvar(:imse | :ipse, maxlag), mcvar.jl:75-105,137-158) need no stored history:
                                    klara_get_chain_acov_mcvar.  Uses a value ring of its own when no history monitor is on. */
    int32_t  sparse_moves;       /* how untuned MH / MALA jobs on the diagonal Gaussian (17 <= D <= 104) keep their running sums.  Two kernel
                                    families run such a job and produce the same bits (both sum in the 8-lane order klara_get_layout reports):
                                    4 lanes per chain, a moving chain's sums folded straight into memory (cheapest while chains move rarely:
                                    acceptance of a few per cent), and 8 lanes per chain with the sums of the chains that moved resident in
                                    registers (flat cost at any acceptance).  0 (default): the library decides launch by launch, on the device,
                                    from the accepted proposals of the previous launch (klara_get_launch_modes reports what ran); 1: always the
                                    4-lane kernels; 2: always the 8-lane kernels.  Results never depend on this field. */

    uint64_t seed;               /* Philox key                                                       */
    uint32_t monitor;            /* KLARA_MON_* bits                                                 */
    int32_t  steps_per_launch;   /* transitions fused in one kernel launch (>=1; 0 = library default) */
    void*    stream;             /* hipStream_t to launch on, or NULL for a library-owned stream     */
} klara_desc;

typedef struct klara_handle klara_handle;

/* Lifetime.  klara_create validates the descriptor the way the Julia constructors do and uploads the
 * target data.  (BasicMCJob inner ctor, BasicMCJob.jl:24-104.) */
klara_status klara_create(const klara_desc* desc, klara_handle** out);
klara_status klara_destroy(klara_handle* h);

/* Initial values (v0 of BasicMCJob.jl:156-185), nchains x ndims row-major.  Evaluates the target and
 * (MALA/HMC) its gradient on device and applies the reference's finiteness asserts
 * (MH.jl:72-85, MALA.jl:76-90, HMC.jl:106-120, SliceSampler.jl:40-48).  Also (re)initialises the tuner
 * state (samplers.jl:29-45: accepted=proposed=0, totproposed=period) and the step counter. */
klara_status klara_set_state(klara_handle* h, const double* x_host);
/* Same, x0 ~ N(0, I) drawn on device from the handle's Philox stream (transition index -1). */
klara_status klara_init_state_normal(klara_handle* h);

/* The transition loop: nsteps calls of iterate! for every chain (BasicMCJob.jl:219-238), including the
 * tuning block and the save rule.  Synchronous w.r.t. the host on return. */
klara_status klara_run(klara_handle* h, int64_t nsteps);
/* Same, but returns immediately after enqueueing (for overlap / external event timing). */
klara_status klara_run_async(klara_handle* h, int64_t nsteps);
klara_status klara_synchronize(klara_handle* h);

/* reset(job[, x]) of BasicMCJob.jl:187-201: rewind sampler/tuner state and counters; x_host may be
 * NULL (keep the current values, re-evaluate the target).  In the reference the random generator keeps advancing across a
 * reset, so run -> reset -> run gives an independent replicate; here the transition index restarts at 0 and the job moves to a
 * fresh Philox key instead: after the k-th klara_reset the key is seed + k * KLARA_EPOCH_KEY_STRIDE (mod 2^64).
 * klara_set_state does not change the key: it replays the job from the given values. */
#define KLARA_EPOCH_KEY_STRIDE 0x9E3779B97F4A7C15ull
klara_status klara_reset(klara_handle* h, const double* x_host);
/* the Philox key the job currently draws from and the number of klara_reset calls so far (either pointer may be NULL) */
klara_status klara_stream_key(klara_handle* h, uint64_t* key, uint64_t* epoch);

/* Read-back.  Any pointer may be NULL. */
klara_status klara_get_state(klara_handle* h, double* x, double* logtarget, double* gradlogtarget);
/* accept diagnostics of the transitions run so far since set_state/reset: steps x nchains bytes,
 * step-major (requires KLARA_MON_ACCEPT).  *nsteps_out receives the number of recorded steps. */
klara_status klara_get_accept_mask(klara_handle* h, uint8_t* mask, int64_t capacity_steps,
                                   int64_t* nsteps_out);
/* the same for the transitions [first_step, first_step + nsteps) only (0-based, within the steps run so far): nsteps x nchains
 * bytes — what a sink that drains a long job chunk by chunk reads (diagnosticvalues of the newly saved steps,
 * BasicContParamIOStream.jl:152-159) instead of every row since the start. */
klara_status klara_get_accept_rows(klara_handle* h, int64_t first_step, int64_t nsteps, uint8_t* mask);
/* per-chain accepted-transition counts over all steps since set_state/reset */
klara_status klara_get_accept_counts(klara_handle* h, uint64_t* naccept, uint64_t* nsteps_out);
/* per-chain sums over saved (postrange) steps: sum[c*D+d], sumsq[c*D+d]; *nsaved_out = count */
klara_status klara_get_chain_sums(klara_handle* h, double* sum, double* sumsq, int64_t* nsaved_out);
/* sums pooled over this handle's chains (reduced on device): sum[D], sumsq[D], total accepted,
 * total transitions, saved samples per chain. */
klara_status klara_get_pooled_summaries(klara_handle* h, double* sum, double* sumsq,
                                        uint64_t* naccept, uint64_t* ntransitions,
                                        int64_t* nsaved_out);
/* ---- multi-GPU without a host framework (SURVEY section 8(b),(e)): one process (or thread) per GPU, chains sharded by
 * klara_desc.chain_offset, and ONE exchange: the all-reduce of the pooled chain summaries over RCCL / xGMI.  (The Python host
 * does the same through torch.distributed, klara.jl_amd/distributed.py.)  librccl.so is loaded on first use (dlopen), a
 * single-GPU user never touches it.  The 128-byte id is RCCL's ncclUniqueId: rank 0 creates it, the caller ships it to the
 * other ranks by whatever it has (file, socket, MPI, Julia Distributed). */
#define KLARA_COMM_ID_BYTES 128
typedef struct klara_comm klara_comm;
klara_status klara_comm_unique_id(uint8_t id[KLARA_COMM_ID_BYTES]);
klara_status klara_comm_init(klara_comm** out, int32_t nranks, int32_t rank, const uint8_t id[KLARA_COMM_ID_BYTES],
                             int32_t device);
/* ranks and this rank's index as the communicator reports them (ncclCommCount, ncclCommUserRank) and the device it was made on:
 * what a launcher checks after the id went round (bench.py's `rccl_ranks_seen`) */
klara_status klara_comm_info(klara_comm* comm, int32_t* nranks, int32_t* rank, int32_t* device);
klara_status klara_comm_destroy(klara_comm* comm);
/* Sum over all ranks of klara_get_pooled_summaries: sum[D], sumsq[D] (NULL unless KLARA_MON_SUMMARIES is on), accepted
 * transitions, transitions, saved samples (= saved steps x chains), chains.  Collective: every rank calls it. */
klara_status klara_gather_summaries(klara_handle* h, klara_comm* comm, double* sum, double* sumsq, uint64_t* naccept,
                                    uint64_t* ntransitions, uint64_t* nsamples, uint64_t* nchains);

/* The pooled posterior moments over every chain of every rank WITHOUT the cancellation of sumsq/n - mean^2 (which loses
 * mean^2/var digits: 4 at the rats model's alpha_c): mean[D], m2[D] = sum (x - mean)^2 per dimension over all saved samples
 * (variance = m2 / nsamples, Klara's var(chain) over the pooled chains), the counters as klara_gather_summaries.  Every chain's
 * running sums become (n, mean, M2) on the device with the one cancelling subtraction carried in double-double, then chains,
 * blocks and ranks are merged by Chan's update; across ranks that is three all-reduces (4 counters, D weighted means, D sums of
 * squares).  comm = NULL: this handle's chains only, no RCCL involved (what a host with its own transport — torch.distributed,
 * MPI.jl, Julia's Distributed — merges itself, klara.jl_amd/distributed.py allreduce_moments).  Requires KLARA_MON_SUMMARIES.
 * Collective when comm != NULL: every rank calls it. */
klara_status klara_gather_moments(klara_handle* h, klara_comm* comm, double* mean, double* m2, uint64_t* nsamples,
                                  uint64_t* naccept, uint64_t* ntransitions, uint64_t* nchains);

/* Memory-safety aid (tests only).  With KLARA_DEBUG_CANARY=1 in the environment every device array the library allocates lies between
 * two 4 KiB canaries of a signalling-NaN pattern; klara_destroy returns KLARA_ERR_STATE when a kernel of the job wrote into one, and this
 * call checks the canaries of everything alive: *nalloc = arrays checked, *ncorrupt = arrays whose canaries were damaged (they are
 * repaired, so a damage is reported once).  poke != 0 first stores 8 bytes right behind the largest live array — the proof that the
 * check fires.  KLARA_ERR_STATE when the canaries are not enabled. */
klara_status klara_selftest_canary(int32_t poke, int64_t* nalloc, int64_t* ncorrupt);

/* one chain of the stored history in Klara's NState layout: value[d + D*i], i = saved step
 * (BasicContMuvParameterNState.jl:89-119); requires KLARA_MON_HISTORY.  With klara_desc.hist_ring_cols > 0 the columns are the
 * last min(saved, hist_ring_cols) saved steps, oldest first, and *ncols_out is that count (klara_get_chain_fields and
 * klara_get_chain_likelihood_prior likewise); klara_saved_steps tells how many steps have been saved in all. */
klara_status klara_get_chain(klara_handle* h, int64_t local_chain, double* value, int64_t capacity_cols,
                             int64_t* ncols_out);
/* the other monitored NState fields of one chain (BasicContMuvParameterNState.jl:1-21): logtarget[i] (n values,
 * KLARA_MON_HIST_LT) and gradlogtarget[d + D*i] (KLARA_MON_HIST_GRAD); either pointer may be NULL. */
klara_status klara_get_chain_fields(klara_handle* h, int64_t local_chain, double* logtarget,
                                    double* gradlogtarget, int64_t capacity_cols, int64_t* ncols_out);
/* loglikelihood[i] and logprior[i] of one chain over the saved steps (KLARA_MON_HIST_LLLP); either pointer may be NULL */
klara_status klara_get_chain_likelihood_prior(klara_handle* h, int64_t local_chain, double* loglikelihood, double* logprior,
                                              int64_t capacity_cols, int64_t* ncols_out);
/* Monte Carlo variance of every (chain, dimension) series of the stored history, computed on device
 * (src/stats/variance/mcvar.jl): iid (:5), batch means with `batchlen` (:35-41), Geyer's initial monotone sequence
 * estimator up to `maxlag` (<= 0: n-1) (:75-105).  Each output is nchains x ndims or NULL; requires KLARA_MON_HISTORY.
 * ess = n * iid / imse and iact = imse / iid (src/stats/convergence/{ess,iact}.jl:3) follow on the host. */
klara_status klara_get_chain_mcvar(klara_handle* h, int64_t batchlen, int64_t maxlag, double* mcvar_iid,
                                   double* mcvar_bm, double* mcvar_imse);
/* Streaming form of mcvar(v, Val{:imse}, maxlag) / mcvar(v, Val{:ipse}, maxlag) (mcvar.jl:75-105, 137-158) for klara_desc.acov_maxlag > 0
 * (maxlag = acov_maxlag): the empirical autocovariances autocov(v, 0:maxlag) (StatsBase, demean = true) of every series are
 * formed exactly from the lagged cross-products, the first and the last maxlag samples and the total, all kept while sampling
 * (3 (maxlag + 1) doubles per series), then Geyer's truncation.  Each output is nchains x ndims or NULL.  Also available post hoc
 * from a stored history: klara_get_chain_mcvar (imse) and klara_get_chain_mcvar_ipse. */
klara_status klara_get_chain_acov_mcvar(klara_handle* h, double* mcvar_imse, double* mcvar_ipse, int64_t* nsamples_out);
/* Geyer's initial positive sequence estimator over the stored history (mcvar.jl:137-158), maxlag <= 0: n - 1 */
klara_status klara_get_chain_mcvar_ipse(klara_handle* h, int64_t maxlag, double* mcvar_ipse);
/* Streaming form of mcvar(v, Val{:bm}) (src/stats/variance/mcvar.jl:35-41: batchlen * var(batch means) / (nbatches *
 * batchlen)) for klara_desc.bm_batchlen > 0: batch means are formed from the running sums at every batch boundary and
 * their mean / sum of squared deviations are updated in place (Welford), so no history is stored — 3 x nchains x ndims
 * doubles however long the run.  mcvar_bm is nchains x ndims (NaN while fewer than two batches are closed). */
klara_status klara_get_chain_bm(klara_handle* h, double* mcvar_bm, int64_t* nbatches_out);
/* saved (post-burn-in, thinned) steps so far */
klara_status klara_saved_steps(klara_handle* h, int64_t* nsaved_out);
/* tuner state per chain (tuners.jl:5-10). In pooled mode every chain reports the shared state. */
klara_status klara_get_tune(klara_handle* h, double* step, int64_t* accepted, int64_t* proposed,
                            int64_t* totproposed);
/* dual-averaging state per chain (DualAveragingMCTune: eps_bar, h_bar); KLARA_TUNER_DUAL_AVERAGING only */
klara_status klara_get_dual_averaging(klara_handle* h, double* epsbar, double* hbar);

/* Measurement: duration (ms, HIP events on the launch stream) and launch count of the transition
 * kernels enqueued by the last klara_run / klara_run_async (after synchronisation). */
klara_status klara_last_run_ms(klara_handle* h, double* kernel_ms, int64_t* nlaunches);

/* Raw device pointers of the state (for zero-copy consumers on the same device, e.g. a torch tensor
 * wrapper or an RCCL gather): x, logtarget, gradlogtarget.  Library keeps ownership. */
klara_status klara_device_ptrs(klara_handle* h, void** x, void** logtarget, void** gradlogtarget);

/* The lane layout the kernels use for this handle (needed by the CPU oracle to sum in the same
 * order): kind 0 = contiguous E elements per lane over G lanes; kind 1 = MFMA-transposed layout
 * (element i on lane-quarter i%4); kind 2 = logistic row split (every lane holds all E elements, the data
 * rows are dealt round-robin to `lanes_per_chain` lanes); kind 3 = pair-transposed layout of the diagonal Gaussian
 * (element pair P = i/2 on lane P % lanes_per_chain, elements_per_lane/2 pairs per lane; 8, 16, 32 or 64 lanes per chain for D <= 128 / 256 / 512 / 1024); kind 4 = hierarchical target
 * with few lanes per chain (unit r on lane r / (elements_per_lane/2), hyper block replicated); kind 5 = logistic regression on the matrix cores
 * (elements as in kind 1; data row r on lane-quarter r % 4: row sums are lane partials over ascending rows, then (q0 + q1) + (q2 + q3); X p and
 * X' (y - 1/(1+exp(-Xp))) are fma chains over ascending columns / rows); kind 6 = dense Gaussian beyond D = 256 on a workgroup of `lanes_per_chain`
 * WAVEFRONTS per tile of 16 chains (`elems_per_lane` = 16, 24 or 32: the elements of a lane's column a wavefront can own) (element i on lane-quarter i % 4 of the wavefront that owns row tile i / 16 — the ceil(D / 16) tiles dealt evenly,
 * consecutive tiles each, the first ceil(D/16) % W wavefronts one more —; lane partials in ascending order, (q0 + q1) + (q2 + q3) inside a wavefront, then
 * the wavefronts in ascending order).  See DESIGN.md section 3. */
klara_status klara_get_layout(klara_handle* h, int32_t* kind, int32_t* lanes_per_chain,
                              int32_t* elems_per_lane);

/* Registers, scratch and static LDS of the transition kernel this handle launches for a launch of `nsteps` transitions (1 selects the
 * one-transition-per-launch instantiation where one exists), read from the loaded code object — no launch happens.  which = 0: the
 * kernel a launch runs (for jobs that two kernel families can run, see klara_desc.sparse_moves: the 4-lane one), 1: the 8-lane
 * sibling of such jobs (the same kernel as 0 otherwise).  bench.py checks the committed PMC summaries against these values, so that
 * counters collected on an older build of a kernel are not silently combined with timings of the current one. */
/* Shader clock (MHz) the device ran at during the handle's last launch of a pair-transposed kernel (layout kind 3): one workgroup in
 * the middle of the grid stores the shader-cycle counter and the constant-rate wall clock when it starts and when it ends; 0 when the
 * handle runs other kernels or has not launched yet.  An MI355X clocks between ~1.9 and 2.4 GHz depending on the power the instruction
 * mix draws: a kernel's issue-cycle budget (bench.py roofline) is only comparable with elapsed time at the clock that was actually
 * running.  Synchronises the handle's streams. */
klara_status klara_get_shader_clock(klara_handle* h, double* mhz);
klara_status klara_get_kernel_attributes(klara_handle* h, int32_t which, int32_t nsteps, int32_t* vgprs, int32_t* scratch_bytes,
                                         int32_t* static_lds_bytes);

/* How the launches of this handle were issued so far (layout kind 3 jobs that both kernel families can run, see
 * klara_desc.sparse_moves; zeros otherwise): counts[0] = launches issued as the 4-lane kernel alone, counts[1] = as the 8-lane
 * kernel alone, counts[2] = as a device-decided pair; last_mode[j] / last_accepted[j] = decision (0: 4 lanes, 1: 8 lanes) and
 * accepted proposals the most recent completed launch of chain partition j < 4 left behind (-1 when not available).  Never
 * synchronises: the per-partition values are read from host-visible memory and may lag the device. */
klara_status klara_get_launch_modes(klara_handle* h, int64_t counts[3], int32_t last_mode[4], int64_t last_accepted[4]);

/* Self-test hook: writes nblocks Philox4x32-10 blocks produced by rocRAND's device engine
 * (rocrand_device::philox4x32_10_engine, seed/subsequence/offset = 4*first_block) into out[4*nblocks];
 * tests compare this with the library's own in-kernel generator and with the CPU oracle. */
klara_status klara_selftest_rocrand_blocks(int32_t device, uint64_t seed, uint64_t subsequence,
                                           uint64_t first_block, int32_t nblocks, uint32_t* out);
/* Self-test hook: evaluates the deterministic device math (log, exp, sincos2pi, normal pair) on n
 * inputs so tests can compare bits with the CPU build of the same header. op: 0 log, 1 exp,
 * 2 sin2pi, 3 cos2pi, 4 sqrt, 5 div (in[i] / in2[i]), 6 erf, 7 log_u01 (log of a positive normal number),
 * 8 sqrt of a Box-Muller radicand (positive normal numbers in [1e-16, 1e2]). */
klara_status klara_selftest_math(int32_t device, int32_t op, int64_t n, const double* in,
                                 const double* in2, double* out);

/* Self-test hook: the proposal normals exactly as the transition kernels draw them — kd_normal_pair_w on both halves of the stream block
 * (seed, chain first_chain + i, transition t, slot 0), i.e. the pair indices 0 and 8 of a transition — counted on the device: counts[k] = number of the
 * 4 * nchains * ntransitions normals with |z| > thr[k] (nthr <= 8); moments (may be NULL) = sum z, sum z^2, sum z^4, max |z|.
 * Tests compare the counts with the CPU build of the same generator (exactly) and with the normal tail mass (statistically). */
klara_status klara_selftest_normal_tail(int32_t device, uint64_t seed, uint64_t first_chain, int64_t nchains,
                                        int64_t ntransitions, int32_t nthr, const double* thr, uint64_t* counts,
                                        double* moments);

/* Self-test hook: the D proposal normals of transition `transition` of chains first_chain .. first_chain + nchains - 1 exactly as the samplers draw
 * them (kd_normal_pair_at: element pair p <- half (p >> 3) & 1 of block slot (p & 7) + 8 (p >> 4)), z[nchains x ndims] row-major, and the
 * transition's accept uniform (block slot ceil(ndims / 2)), accept_u[nchains] (may be NULL).  The joint-law tests of the stream — chi-square of
 * a transition's sum z^2, independence of the two pairs that share a block and of a pair's radius and angle bits — run on this output, and
 * the CPU build of the same generator (oracle ko_transition_normals) must return the same bits. */
klara_status klara_selftest_transition_normals(int32_t device, uint64_t seed, uint64_t first_chain, int64_t nchains,
                                               uint64_t transition, int32_t ndims, double* z, double* accept_u);

/* Self-test hook: D(16x16) = A(16x4) * B(4x16) + C(16x16), all row-major, through ONE
 * v_mfma_f64_16x16x4_f64 — pins the instruction's accumulation order for the dense-target parity. */
klara_status klara_selftest_mfma_f64(int32_t device, const double* A, const double* B, const double* C,
                                     double* D);

/* One v_mfma_f64_4x4x4_4b with raw per-lane operands (64 doubles each): A_b[i][k] on lane 16k + 4b + i, B_b[k][j] on
 * lane 16k + 4b + j, C/D_b[i][j] on lane 16i + 4b + j.  Pins the lane layout and the accumulation order the dense
 * kernel's 4-row tail tile relies on (klara_dense.h). */
klara_status klara_selftest_mfma_f64_4x4x4(int32_t device, const double* A, const double* B, const double* C,
                                           double* D);

/* Self-test hook, no device needed: the launches a sequence of klara_run calls of the given lengths issues on a fresh job of
 * this descriptor — transitions per launch k[i], the save-rule bookkeeping handed to the kernels (columns already saved,
 * thinning phase of the launch's first post-burn-in transition) and flags (bit 0: a pooled tuner update follows, bit 1: a batch
 * of the streaming batch means closes, bit 2: last launch of its klara_run call).  Launches end after steps_per_launch
 * transitions, at the pooled tuner's events (tuners.jl:27-32) and at batch boundaries, whichever comes first. */
klara_status klara_selftest_plan(const klara_desc* desc, int32_t nruns, const int64_t* run_lengths, int64_t capacity, int64_t* k,
                                 int64_t* save_col0, int32_t* save_phase0, int32_t* flags, int64_t* nlaunches);

/* CUSTOM target: compile `src` for gfx950 exactly as klara_create would for this sampler and dimension, without creating
 * a handle and without needing a GPU (a user checks a closure before submitting a job).  KLARA_OK or KLARA_ERR_COMPILE. */
klara_status klara_check_custom_target(const char* src, int32_t sampler, int32_t ndims);
/* Compiler output of the calling thread's last klara_create / klara_check_custom_target that compiled a CUSTOM target
 * ("" if none); valid until the thread's next such call. */
const char* klara_compile_log(void);

const char* klara_strerror(klara_status s);
int32_t klara_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* KLARA_HIP_H */
