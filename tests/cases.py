"""Shared parity cases: the same configuration dict drives the product Engine (GPU) and the CPU oracle."""
from __future__ import annotations

from pathlib import Path

import numpy as np

import klara_jl_amd as K
from klara_jl_amd import _lib as L

ROOT = Path(__file__).resolve().parent.parent
GOLDEN = ROOT / "tests" / "golden"


def swiss_data():
    """Swiss banknote design matrix, standardised as doc/examples/swiss/MALA/analytical.jl:3-9 does
    ((x - mean) / std with the n-1 sample std), from the committed fixture of the reference's data files."""
    raw = np.load(GOLDEN / "swiss.npz")
    X = raw["measurements"]
    X = (X - X.mean(axis=0)) / X.std(axis=0, ddof=1)
    return np.ascontiguousarray(X), np.ascontiguousarray(raw["status"].astype(np.float64))


def rats_target():
    """data/rats/{weight,age}.csv of the reference (fixture tests/golden/rats.npz); ages centred at 22 as in BUGS."""
    raw = np.load(GOLDEN / "rats.npz")
    return K.HierNormalTarget(raw["weight"], raw["age"] - 22.0)


def compound_symmetric_precision(d, rho=0.5):
    return (np.eye(d) - (rho / (1.0 - rho + d * rho)) * np.ones((d, d))) / (1.0 - rho)


# ---- likelihood + prior closures: the Normal-Normal model of the reference's parameter tests
# (/root/reference/test/BasicContMuvParameter.jl:232-292, 326-453, 458-530): x | mu ~ N(mu, diag(s)), mu ~ N(mu0, diag(s0)); the
# parameter is mu, the data block is (x[D], s[D], mu0[D], s0[D]).  Written with diagonal covariances (the reference's test values
# are eye / diagm): logpdf = -1/2 (sum d_i^2 / s_i + D log 2 pi + sum log s_i).
SRC_NN_LL = r"""
KLARA_USER_FN double klara_user_loglikelihood(const double* mu, int D, const double* data, long long ndata)
{
    const double* x = data; const double* s = data + KLARA_D;
    double q = 0.0, ld = 0.0;
    for (int i = 0; i < KLARA_D; ++i) { const double d = x[i] - mu[i]; q = q + d * d / s[i]; ld = ld + kd_log(s[i]); }
    return -0.5 * (q + (double)KLARA_D * 1.8378770664093453 + ld);
}
"""
SRC_NN_LP = r"""
KLARA_USER_FN double klara_user_logprior(const double* mu, int D, const double* data, long long ndata)
{
    const double* m0 = data + 2 * KLARA_D; const double* s0 = data + 3 * KLARA_D;
    double q = 0.0, ld = 0.0;
    for (int i = 0; i < KLARA_D; ++i) { const double d = mu[i] - m0[i]; q = q + d * d / s0[i]; ld = ld + kd_log(s0[i]); }
    return -0.5 * (q + (double)KLARA_D * 1.8378770664093453 + ld);
}
"""
SRC_NN_GLL = r"""
KLARA_USER_FN void klara_user_gradloglikelihood(const double* mu, int D, const double* data, long long ndata, double* g)
{
    const double* x = data; const double* s = data + KLARA_D;
    for (int i = 0; i < KLARA_D; ++i) g[i] = (x[i] - mu[i]) / s[i];
}
"""
SRC_NN_GLP = r"""
KLARA_USER_FN void klara_user_gradlogprior(const double* mu, int D, const double* data, long long ndata, double* g)
{
    const double* m0 = data + 2 * KLARA_D; const double* s0 = data + 3 * KLARA_D;
    for (int i = 0; i < KLARA_D; ++i) g[i] = -(mu[i] - m0[i]) / s0[i];
}
"""


def normal_normal_target(x, s, mu0, s0):
    """CustomTarget in likelihood + prior form for the Normal-Normal model above."""
    d = len(x)
    return K.CustomTarget.likelihood_prior(d, SRC_NN_LL, SRC_NN_LP, SRC_NN_GLL, SRC_NN_GLP, np.concatenate([x, s, mu0, s0]).astype(np.float64))


# ---- user-defined targets (KLARA_TARGET_CUSTOM): C text compiled by hiprtc for the device and by gcc for the oracle
SRC_NEGDOT = r"""
/* README.md:23,155: plogtarget(z) = -dot(z, z), pgradlogtarget(z) = -2*z */
KLARA_USER_FN double klara_user_logtarget(const double* x, int D, const double* data, long long ndata)
{
    double s = 0.0;
    for (int i = 0; i < KLARA_D; ++i) s = s + x[i] * x[i];
    return 0.0 - s;
}
KLARA_USER_FN void klara_user_gradlogtarget(const double* x, int D, const double* data, long long ndata, double* g)
{
    for (int i = 0; i < KLARA_D; ++i) g[i] = -2.0 * x[i];
}
"""

SRC_BANANA = r"""
/* a curved two-dimensional density: lt = -(1 - x0)^2 / 20 - (x1 - x0^2)^2 */
KLARA_USER_FN double klara_user_logtarget(const double* x, int D, const double* data, long long ndata)
{
    const double a = 1.0 - x[0], b = x[1] - x[0] * x[0];
    return -(a * a) / 20.0 - b * b;
}
KLARA_USER_FN void klara_user_gradlogtarget(const double* x, int D, const double* data, long long ndata, double* g)
{
    const double a = 1.0 - x[0], b = x[1] - x[0] * x[0];
    g[0] = a / 10.0 + 4.0 * (b * x[0]);
    g[1] = -2.0 * b;
}
"""

SRC_BANANA_LT_ONLY = SRC_BANANA[:SRC_BANANA.index("KLARA_USER_FN void")]     # MH / slice need no gradient closure

SRC_LOGIT = r"""
/* doc/examples/swiss/MALA/analytical.jl:11-18 written as a user closure; data = [X (n x D row-major), y (n), lambda] */
KLARA_USER_FN double klara_user_logtarget(const double* p, int D, const double* data, long long ndata)
{
    const int n = (int)((ndata - 1) / (KLARA_D + 1));
    const double* X = data; const double* y = data + (long long)n * KLARA_D; const double lambda = data[ndata - 1];
    double dotxy = 0.0, slog = 0.0;
    for (int r = 0; r < n; ++r) {
        double xp = 0.0;
        for (int e = 0; e < KLARA_D; ++e) xp = kd_fma(X[r * KLARA_D + e], p[e], xp);
        dotxy = dotxy + xp * y[r];
        double sp, lg;
        kd_softplus_logistic_rows(xp, &sp, &lg);
        slog = slog + sp;
    }
    double dotpp = 0.0;
    for (int e = 0; e < KLARA_D; ++e) dotpp = dotpp + p[e] * p[e];
    return (dotxy - slog) + -0.5 * (dotpp / lambda + (double)KLARA_D * kd_log(2.0 * 3.141592653589793 * lambda));
}
KLARA_USER_FN void klara_user_gradlogtarget(const double* p, int D, const double* data, long long ndata, double* g)
{
    const int n = (int)((ndata - 1) / (KLARA_D + 1));
    const double* X = data; const double* y = data + (long long)n * KLARA_D; const double lambda = data[ndata - 1];
    for (int e = 0; e < KLARA_D; ++e) g[e] = 0.0;
    for (int r = 0; r < n; ++r) {
        double xp = 0.0;
        for (int e = 0; e < KLARA_D; ++e) xp = kd_fma(X[r * KLARA_D + e], p[e], xp);
        double sp, lg;
        kd_softplus_logistic_rows(xp, &sp, &lg);
        const double res = y[r] - lg;
        for (int e = 0; e < KLARA_D; ++e) g[e] = kd_fma(X[r * KLARA_D + e], res, g[e]);
    }
    for (int e = 0; e < KLARA_D; ++e) g[e] = g[e] - p[e] / lambda;
}
"""

SRC_QUARTIC_CHAIN = r"""
/* non-Gaussian, coupled: lt = -sum_i (x_i^2 / 2 + c x_i^4) - k/2 sum_i (x_{i+1} - x_i)^2, data = [c, k] */
KLARA_USER_FN double klara_user_logtarget(const double* x, int D, const double* data, long long ndata)
{
    const double c = data[0], k = data[1];
    double s = 0.0;
    for (int i = 0; i < KLARA_D; ++i) { const double q = x[i] * x[i]; s = s + (0.5 * q + c * (q * q)); }
    double t = 0.0;
    for (int i = 0; i + 1 < KLARA_D; ++i) { const double d = x[i + 1] - x[i]; t = t + d * d; }
    return -s - (0.5 * k) * t;
}
KLARA_USER_FN void klara_user_gradlogtarget(const double* x, int D, const double* data, long long ndata, double* g)
{
    const double c = data[0], k = data[1];
    for (int i = 0; i < KLARA_D; ++i) {
        const double q = x[i] * x[i];
        double v = -(x[i] + (4.0 * c) * (q * x[i]));
        if (i + 1 < KLARA_D) v = v + k * (x[i + 1] - x[i]);
        if (i > 0) v = v - k * (x[i] - x[i - 1]);
        g[i] = v;
    }
}
"""


# ---- pair closures (K.CustomTarget.pairwise, include/klara_hip.h KLARA_USER_PAIR_TARGET): lt = sum over element pairs
SRC_PAIR_NEGDOT = r"""
/* README.md:23 logtarget = -dot(z, z), :155 gradlogtarget = -2 z, one element pair at a time */
KLARA_USER_FN double klara_user_pair(double x0, double x1, int pair, int D, const double* data, long long ndata, double* g0, double* g1)
{
    *g0 = -2.0 * x0; *g1 = -2.0 * x1;
    return -(x0 * x0 + x1 * x1);
}
"""
SRC_PAIR_INDEXED = r"""
/* lt = -sum_i w_i x_i^2 with w = data[0 .. D): the closure indexes its data block BY COORDINATE (ndata = D exactly), which is only safe
 * because it is called for real pairs alone, pair < ceil(D/2) (ADVICE r3: the kernels used to call it for the padding pairs of a lane too,
 * pair up to NP*Q - 1, and mask the result).  A call outside that range stops the kernel (and the host build) at once. */
KLARA_USER_FN double klara_user_pair(double x0, double x1, int pair, int D, const double* data, long long ndata, double* g0, double* g1)
{
    if (pair < 0 || 2 * pair >= D || 2 * pair >= ndata) __builtin_trap();
    const double w0 = data[2 * pair], w1 = 2 * pair + 1 < D ? data[2 * pair + 1] : 0.0;
    *g0 = -2.0 * (w0 * x0); *g1 = -2.0 * (w1 * x1);
    return -(w0 * (x0 * x0)) - w1 * (x1 * x1);
}
"""
SRC_PAIR_QUARTIC = r"""
/* non-Gaussian, coupled within the pair: -(x0^2/2 + c x0^4) - (x1^2/2 + c x1^4) - k/2 (x1 - x0)^2, data = [c, k]; the half pair of
 * an odd D (2 pair + 1 == D) has no second coordinate */
KLARA_USER_FN double klara_user_pair(double x0, double x1, int pair, int D, const double* data, long long ndata, double* g0, double* g1)
{
    const double c = data[0], k = data[1];
    const double q0 = x0 * x0, q1 = x1 * x1;
    if (2 * pair + 1 >= D) { *g0 = -(x0 + (4.0 * c) * (q0 * x0)); *g1 = 0.0; return -(0.5 * q0 + c * (q0 * q0)); }
    const double d = x1 - x0;
    *g0 = -(x0 + (4.0 * c) * (q0 * x0)) + k * d;
    *g1 = -(x1 + (4.0 * c) * (q1 * x1)) - k * d;
    return -((0.5 * q0 + c * (q0 * q0)) + (0.5 * q1 + c * (q1 * q1))) - (0.5 * k) * (d * d);
}
"""
SRC_PAIR_BANANA = r"""
/* the twisted ("banana") Gaussian on every pair: -x0^2 / (2 s) - (x1 + b x0^2 - s b)^2 / 2, data = [b, s] */
KLARA_USER_FN double klara_user_pair(double x0, double x1, int pair, int D, const double* data, long long ndata, double* g0, double* g1)
{
    const double b = data[0], s = data[1];
    const double w = kd_fma(b, x0 * x0, x1) - s * b;
    *g0 = -x0 / s - (w * (2.0 * b)) * x0;
    *g1 = -w;
    return -(x0 * x0) / (2.0 * s) - 0.5 * (w * w);
}
"""


def synthetic_logit(n, d, seed=5):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, d))
    y = (rng.random(n) < 1.0 / (1.0 + np.exp(-(X @ np.linspace(-1.0, 1.0, d))))).astype(np.float64)
    return np.ascontiguousarray(X), y


# (D, data rows, engine settings) of the matrix-core logistic cases
LOGIT_MFMA_CASES = {
    "mala_logitm_d20": (20, 200, dict(sampler=L.SAMPLER_MALA, driftstep=0.35)),
    "hmc_logitm_d33_n70": (33, 70, dict(sampler=L.SAMPLER_HMC, leapstep=0.9, nleaps=4, nchains=35)),                       # odd D: NE = 16; 70 rows: 5 tiles, the last with 6 rows
    "mh_logitm_d64_n300": (64, 300, dict(sampler=L.SAMPLER_MH, mh_sigma=0.03, thinning=2)),
    "mala_logitm_d128_n1100_tuned": (128, 1100, dict(sampler=L.SAMPLER_MALA, driftstep=0.3, tuner=L.TUNER_ACCEPT_RATE, targetrate=0.574, period=5, nchains=19, nsteps=22)),
    "hmc_logitm_d40_dualavg": (40, 131, dict(sampler=L.SAMPLER_HMC, leapstep=0.05, nleaps=4, tuner=L.TUNER_DUAL_AVERAGING, targetrate=0.65, da_nadapt=15)),
    "hmc_logitm_d17_pooled": (17, 16, dict(sampler=L.SAMPLER_HMC, leapstep=1.2, nleaps=3, tuner=L.TUNER_ACCEPT_RATE, tuner_mode=L.TUNE_POOLED, targetrate=0.8, period=5)),   # one tile of rows exactly
    "mala_logitm_d96_n17_verbose": (96, 17, dict(sampler=L.SAMPLER_MALA, driftstep=1.5, verbose=True, period=10)),         # NE = 24; a second tile with one row
    "mh_logitm_d100_n33": (100, 33, dict(sampler=L.SAMPLER_MH, mh_sigma=0.2, nchains=70)),
    "mala_logitm_d200_n40": (200, 40, dict(sampler=L.SAMPLER_MALA, driftstep=0.8, nchains=19, nsteps=14)),                   # NE = 56: one wavefront per SIMD
    "hmc_logitm_d256_n65_tuned": (256, 65, dict(sampler=L.SAMPLER_HMC, leapstep=0.9, nleaps=3, tuner=L.TUNER_ACCEPT_RATE, targetrate=0.7, period=5, nchains=19, nsteps=14)),
    "slice_logitm_d20": (20, 50, dict(sampler=L.SAMPLER_SLICE, slice_widths=1.5, nchains=35, nsteps=8, burnin=2)),         # a probe = pass 1 + the rows; chains of a tile out of lockstep
    "slice_logitm_d40_nostepout": (40, 17, dict(sampler=L.SAMPLER_SLICE, slice_widths=2.5, slice_stepout=False, nchains=19, nsteps=6, burnin=1)),
}


def _split_cases():
    return {
        "hmc_dense_d300_split": (300, dict(sampler=L.SAMPLER_HMC, nsteps=10, burnin=2, leapstep=0.08, nleaps=3)),
        "hmc_dense_d257_split_pooled": (257, dict(sampler=L.SAMPLER_HMC, nsteps=36, burnin=24, leapstep=0.12, nleaps=3, tuner=L.TUNER_ACCEPT_RATE,
                                                  tuner_mode=L.TUNE_POOLED, targetrate=0.65, period=8, nchains=37)),
        "mala_dense_d512_split_mean_tuned": (512, dict(sampler=L.SAMPLER_MALA, nsteps=30, burnin=15, driftstep=0.03, tuner=L.TUNER_ACCEPT_RATE,
                                                       targetrate=0.574, period=5)),
        "mh_dense_d700_split_mean": (700, dict(sampler=L.SAMPLER_MH, nsteps=24, burnin=4, thinning=2, nchains=19)),
        "hmc_dense_d1024_split_dualavg": (1024, dict(sampler=L.SAMPLER_HMC, nsteps=12, burnin=0, leapstep=0.06, nleaps=3, tuner=L.TUNER_DUAL_AVERAGING,
                                                     targetrate=0.8, da_nadapt=7, nchains=18)),
        "mala_dense_d1000_split_pooled": (1000, dict(sampler=L.SAMPLER_MALA, nsteps=30, burnin=20, driftstep=0.02, tuner=L.TUNER_ACCEPT_RATE,
                                                     tuner_mode=L.TUNE_POOLED, targetrate=0.574, period=10, nchains=33)),
        "mh_dense_d333_split": (333, dict(sampler=L.SAMPLER_MH, nsteps=20, burnin=0, nchains=5)),
        "slice_dense_d260_split": (260, dict(sampler=L.SAMPLER_SLICE, nsteps=3, burnin=0, nchains=19)),
        "slice_dense_d530_split_mean": (530, dict(sampler=L.SAMPLER_SLICE, nsteps=2, burnin=0, nchains=7, slice_stepout=False)),
    }


SPLIT_CASES = {k: (v[0], dict(v[1])) for k, v in _split_cases().items()}


def make_case(name):
    """name -> dict(engine kwargs..., target=<family object>, x0=None|array)."""
    c = {}
    if name == "mh_readme":            # BASELINE cfg 1 (README.md:23-50), replicated over 8 chains
        c = dict(sampler=L.SAMPLER_MH, target=K.GaussDiagTarget.negdot(2), nchains=8, nsteps=400, burnin=100,
                 mh_sigma=[1.0, 1.0], x0=np.tile([5.1, -0.9], (8, 1)))
    elif name == "mh_d100":
        c = dict(sampler=L.SAMPLER_MH, target=K.GaussDiagTarget.negdot(100), nchains=70, nsteps=60, burnin=10,
                 mh_sigma=np.full(100, 0.1))
    elif name == "mh_mvnormal_d7":     # odd D, per-dim weights and means
        mu = np.linspace(-2, 3, 7); sg = np.linspace(0.5, 2.0, 7)
        c = dict(sampler=L.SAMPLER_MH, target=K.GaussDiagTarget.mvnormal(mu, sg), nchains=33, nsteps=80, burnin=0,
                 mh_sigma=sg * 0.8)
    elif name == "mala_d100":          # BASELINE cfg 2 shape at parity-test size
        c = dict(sampler=L.SAMPLER_MALA, target=K.GaussDiagTarget.negdot(100), nchains=130, nsteps=50, burnin=10,
                 driftstep=0.9)
    elif name == "mala_d100_small_step":
        c = dict(sampler=L.SAMPLER_MALA, target=K.GaussDiagTarget.negdot(100), nchains=66, nsteps=60, burnin=20,
                 thinning=3, driftstep=0.05)
    elif name == "mala_d3_tuned":      # AcceptanceRate tuner, per chain, several tuning events
        c = dict(sampler=L.SAMPLER_MALA, target=K.GaussDiagTarget.negdot(3), nchains=50, nsteps=260, burnin=200,
                 driftstep=1.5, tuner=L.TUNER_ACCEPT_RATE, targetrate=0.574, period=25)
    elif name == "mala_d20_tuned":     # AcceptanceRate per chain on an even-D diagonal target (pair-transposed layout)
        c = dict(sampler=L.SAMPLER_MALA, target=K.GaussDiagTarget.negdot(20), nchains=43, nsteps=260, burnin=200,
                 driftstep=1.2, tuner=L.TUNER_ACCEPT_RATE, targetrate=0.574, period=25)
    elif name == "hmc_d100_tuned":     # AcceptanceRate per chain, HMC: the step enters every leapfrog
        c = dict(sampler=L.SAMPLER_HMC, target=K.GaussDiagTarget.mvnormal(np.linspace(-1, 1, 100), np.linspace(0.5, 1.5, 100)),
                 nchains=21, nsteps=90, burnin=60, leapstep=0.25, nleaps=6, tuner=L.TUNER_ACCEPT_RATE, targetrate=0.8, period=15)
    elif name == "mala_d100_verbose":  # VanillaMCTuner(verbose=true): proposals are counted, nothing tunes
        c = dict(sampler=L.SAMPLER_MALA, target=K.GaussDiagTarget.negdot(100), nchains=20, nsteps=70, burnin=30,
                 driftstep=0.05, verbose=True, period=20)
    elif name == "mala_d1":            # the univariate case is the D-vector kernel with D = 1
        c = dict(sampler=L.SAMPLER_MALA, target=K.GaussDiagTarget.negdot(1), nchains=100, nsteps=60, burnin=10, driftstep=0.8)
    elif name == "hmc_d128_full":      # no padding lane: G*E == D, so the accept uniform takes the explicit path
        c = dict(sampler=L.SAMPLER_HMC, target=K.GaussDiagTarget.negdot(128), nchains=21, nsteps=15, burnin=0,
                 leapstep=0.08, nleaps=5)
    elif name == "mala_d129":          # E=4 on all 64 lanes (33 busy), one chain per wavefront
        c = dict(sampler=L.SAMPLER_MALA, target=K.GaussDiagTarget.negdot(129), nchains=7, nsteps=20, burnin=0, driftstep=0.03)
    elif name == "mh_d512":            # E=8, the largest supported diagonal target
        c = dict(sampler=L.SAMPLER_MH, target=K.GaussDiagTarget.negdot(512), nchains=5, nsteps=25, burnin=0,
                 mh_sigma=np.full(512, 0.03))
    elif name == "hmc_d1000_tuned":    # round 6: 64 lanes per chain (513 <= D <= 1024: one chain per wavefront), AcceptanceRate per chain
        c = dict(sampler=L.SAMPLER_HMC, target=K.GaussDiagTarget.mvnormal(np.linspace(-1, 1, 1000), np.linspace(0.5, 1.5, 1000)), nchains=5, nsteps=40, burnin=30,
                 leapstep=0.12, nleaps=4, tuner=L.TUNER_ACCEPT_RATE, targetrate=0.7, period=10)
    elif name == "mala_d777_pooled":   # odd D (a half pair), pooled tuner
        c = dict(sampler=L.SAMPLER_MALA, target=K.GaussDiagTarget.negdot(777), nchains=7, nsteps=50, burnin=40, thinning=3, driftstep=0.02,
                 tuner=L.TUNER_ACCEPT_RATE, tuner_mode=L.TUNE_POOLED, targetrate=0.574, period=10)
    elif name == "slice_d600":         # the free-running slice kernel on 64 lanes per chain
        c = dict(sampler=L.SAMPLER_SLICE, target=K.GaussDiagTarget.mvnormal(np.linspace(-1, 1, 600), np.linspace(0.7, 1.5, 600)), nchains=3, nsteps=5, burnin=1,
                 slice_widths=np.linspace(0.5, 2.5, 600))
    elif name == "pair_quartic_hmc_d700_dualavg":      # a pair closure at 64 lanes per chain
        c = dict(sampler=L.SAMPLER_HMC, target=K.CustomTarget.pairwise(700, SRC_PAIR_QUARTIC, [0.05, 0.3]), nchains=5, nsteps=30, burnin=5,
                 leapstep=0.05, nleaps=3, tuner=L.TUNER_DUAL_AVERAGING, targetrate=0.65, da_nadapt=15, x0=0.4 * np.random.default_rng(8).standard_normal((5, 700)))
    elif name == "slice_d2_mvnormal":  # G=1: one chain per lane with divergent step-out / shrink loops
        c = dict(sampler=L.SAMPLER_SLICE, target=K.GaussDiagTarget.mvnormal([1.0, -2.0], [0.5, 3.0]), nchains=130,
                 nsteps=25, burnin=5, slice_widths=[0.3, 4.0])
    elif name == "mala_d3_tuned_erf":  # erf_rate_score(x, 3) instead of the logistic score
        c = dict(sampler=L.SAMPLER_MALA, target=K.GaussDiagTarget.negdot(3), nchains=50, nsteps=260, burnin=200,
                 driftstep=1.5, tuner=L.TUNER_ACCEPT_RATE, targetrate=0.574, period=25, tuner_score=1, score_k=3.0)
    elif name == "mala_d300":          # E=4 layout
        c = dict(sampler=L.SAMPLER_MALA, target=K.GaussDiagTarget.negdot(300), nchains=9, nsteps=20, burnin=0,
                 driftstep=0.02)
    elif name == "hmc_d100":           # north-star sampler on the README target
        c = dict(sampler=L.SAMPLER_HMC, target=K.GaussDiagTarget.negdot(100), nchains=70, nsteps=30, burnin=5,
                 leapstep=0.1, nleaps=10)
    elif name == "hmc_d10_tuned_pooled":
        c = dict(sampler=L.SAMPLER_HMC, target=K.GaussDiagTarget.negdot(10), nchains=40, nsteps=120, burnin=90,
                 leapstep=0.9, nleaps=5, tuner=L.TUNER_ACCEPT_RATE, tuner_mode=L.TUNE_POOLED, targetrate=0.65,
                 period=20)
    elif name == "hmc_d10_dualavg":    # DualAveragingMCTuner: per-chain step AND per-chain nleaps (iterate/HMC.jl:142-144)
        c = dict(sampler=L.SAMPLER_HMC, target=K.GaussDiagTarget.negdot(10), nchains=45, nsteps=80, burnin=50,
                 leapstep=0.3, nleaps=6, tuner=L.TUNER_DUAL_AVERAGING, targetrate=0.65, da_nadapt=50)
    elif name == "hmc_d40_dualavg":    # dual averaging on the pair-transposed layout (D >= 18), non-unit diagonal, verbose counting
        c = dict(sampler=L.SAMPLER_HMC, target=K.GaussDiagTarget.mvnormal(np.linspace(-1, 1, 40), np.linspace(0.6, 1.6, 40)), nchains=27,
                 nsteps=70, burnin=40, leapstep=0.25, nleaps=5, tuner=L.TUNER_DUAL_AVERAGING, targetrate=0.65, da_nadapt=40, verbose=True, period=10)
    elif name == "hmc_d100_dualavg":   # unit weights, NP = 7
        c = dict(sampler=L.SAMPLER_HMC, target=K.GaussDiagTarget.negdot(100), nchains=19, nsteps=50, burnin=30,
                 leapstep=0.2, nleaps=4, tuner=L.TUNER_DUAL_AVERAGING, targetrate=0.7, da_nadapt=30)
    elif name == "hmc_dense_d37_dualavg":
        rng = np.random.default_rng(5)
        a = rng.standard_normal((37, 37)); p = a @ a.T / 37 + np.eye(37)
        c = dict(sampler=L.SAMPLER_HMC, target=K.GaussDenseTarget(p), nchains=21, nsteps=40, burnin=0, leapstep=0.2,
                 nleaps=4, tuner=L.TUNER_DUAL_AVERAGING, targetrate=0.8, da_nadapt=25, verbose=True, period=10)
    elif name == "hmc_rats_dualavg":
        t = rats_target()
        x0 = t.least_squares_start()[None, :] + 0.05 * np.random.default_rng(8).standard_normal((18, t.ndims))
        c = dict(sampler=L.SAMPLER_HMC, target=t, nchains=18, nsteps=40, burnin=30, leapstep=0.02, nleaps=5,
                 tuner=L.TUNER_DUAL_AVERAGING, targetrate=0.65, da_nadapt=30, x0=x0)
    elif name == "hmc_dense_d100":     # BASELINE cfg 3 shape at parity-test size (FP64 MFMA path)
        c = dict(sampler=L.SAMPLER_HMC, target=K.GaussDenseTarget(compound_symmetric_precision(100)), nchains=40,
                 nsteps=12, burnin=2, leapstep=0.1, nleaps=10)
    elif name == "hmc_dense_d37":      # ragged D (not a multiple of 4 or 16), random SPD precision
        rng = np.random.default_rng(5)
        a = rng.standard_normal((37, 37)); p = a @ a.T / 37 + np.eye(37)
        c = dict(sampler=L.SAMPLER_HMC, target=K.GaussDenseTarget(p), nchains=19, nsteps=15, burnin=0,
                 leapstep=0.15, nleaps=4)
    elif name in ("hmc_dense_d98", "hmc_dense_d70", "hmc_dense_d128"):   # NE = 25 with 2 of the 4 tail rows / with none; NE = 32
        d = int(name[-3:].lstrip("d")) if name != "hmc_dense_d98" and name != "hmc_dense_d70" else int(name[-2:])
        rng = np.random.default_rng(d)
        a = rng.standard_normal((d, d)); p = a @ a.T / d + np.eye(d)
        c = dict(sampler=L.SAMPLER_HMC, target=K.GaussDenseTarget(p), nchains=21, nsteps=10, burnin=0, leapstep=0.12, nleaps=3)
    elif name in ("hmc_dense_d100_mean", "mala_dense_d37_mean", "mh_dense_d20_mean", "hmc_dense_d70_mean_dualavg"):   # (x - mu)' P (x - mu)
        d = int(name.split("_")[2][1:])
        rng = np.random.default_rng(100 + d)
        a = rng.standard_normal((d, d)); p = a @ a.T / d + np.eye(d)
        t = K.GaussDenseTarget(p, const=0.75, mu=rng.standard_normal(d) * 2.0)
        x0 = t.mu[None, :] + rng.standard_normal((23, d))
        if name.startswith("hmc") and not name.endswith("dualavg"):
            c = dict(sampler=L.SAMPLER_HMC, target=t, nchains=23, nsteps=12, burnin=2, leapstep=0.1, nleaps=5, x0=x0)
        elif name.startswith("hmc"):
            c = dict(sampler=L.SAMPLER_HMC, target=t, nchains=23, nsteps=30, burnin=0, leapstep=0.2, nleaps=4, x0=x0,
                     tuner=L.TUNER_DUAL_AVERAGING, targetrate=0.8, da_nadapt=20)
        elif name.startswith("mala"):
            c = dict(sampler=L.SAMPLER_MALA, target=t, nchains=23, nsteps=25, burnin=5, driftstep=0.2, x0=x0,
                     tuner=L.TUNER_ACCEPT_RATE, targetrate=0.574, period=5)
        else:
            c = dict(sampler=L.SAMPLER_MH, target=t, nchains=23, nsteps=40, burnin=0, mh_sigma=np.full(d, 0.25), x0=x0)
    elif name == "hmc_dense_d130_wide":    # (round 4: HMC at D = 130 runs on the streamed matrix-core layout; the closure form of the dense target is mh_dense_d130_wide's)
        rng = np.random.default_rng(131)
        a = rng.standard_normal((130, 130)); p = a @ a.T / 130 + np.eye(130)
        c = dict(sampler=L.SAMPLER_HMC, target=K.GaussDenseTarget(p, const=-0.5), nchains=9, nsteps=8, burnin=1, leapstep=0.1, nleaps=3,
                 x0=rng.standard_normal((9, 130)))
    elif name in ("hmc_dense_d256_stream_tuned", "hmc_dense_d192_stream_mean", "hmc_dense_d160_stream_pooled", "hmc_dense_d129_stream", "hmc_dense_d201_stream"):
        # HMC beyond D = 128 on the matrix cores: P streamed from memory, momentum in LDS (klara_dense_big.h; NE = 40 / 48 / 56 / 64 elements per lane)
        d = {"hmc_dense_d256_stream_tuned": 256, "hmc_dense_d192_stream_mean": 192, "hmc_dense_d160_stream_pooled": 160, "hmc_dense_d129_stream": 129,
             "hmc_dense_d201_stream": 201}[name]
        rng = np.random.default_rng(d)
        a = rng.standard_normal((d, d)); pm = a @ a.T / d + np.eye(d)
        mu = rng.standard_normal(d) if name in ("hmc_dense_d192_stream_mean", "hmc_dense_d256_stream_tuned") else None
        t = K.GaussDenseTarget(pm, const=0.25, mu=mu)
        x0 = rng.standard_normal((37, d)) + (0.0 if mu is None else mu[None, :])
        kw = {"hmc_dense_d256_stream_tuned": dict(nsteps=40, burnin=20, leapstep=0.12, nleaps=4, tuner=L.TUNER_ACCEPT_RATE, targetrate=0.7, period=5),
              "hmc_dense_d192_stream_mean": dict(nsteps=14, burnin=3, thinning=2, leapstep=0.1, nleaps=5),
              "hmc_dense_d160_stream_pooled": dict(nsteps=40, burnin=30, leapstep=0.15, nleaps=3, tuner=L.TUNER_ACCEPT_RATE, tuner_mode=L.TUNE_POOLED,
                                                   targetrate=0.65, period=10),
              "hmc_dense_d129_stream": dict(nsteps=10, burnin=0, leapstep=0.1, nleaps=2),
              "hmc_dense_d201_stream": dict(nsteps=12, burnin=2, leapstep=0.08, nleaps=3)}[name]
        c = dict(sampler=L.SAMPLER_HMC, target=t, nchains=37, x0=x0, **kw)
    elif name in ("mala_dense_d200_stream_tuned", "mh_dense_d256_stream_mean", "mala_dense_d130_stream_pooled", "hmc_dense_d130_dualavg_wide"):
        # MALA / MH beyond D = 128 on the streamed matrix-core layout (no vector beyond x and P x is held: the current value is re-read from X for the
        # backward term); HMC with dual averaging streams too (per-chain trip counts: the wavefront runs to the longest trajectory of its 16 chains)
        d = {"mala_dense_d200_stream_tuned": 200, "mh_dense_d256_stream_mean": 256, "mala_dense_d130_stream_pooled": 130, "hmc_dense_d130_dualavg_wide": 130}[name]
        rng = np.random.default_rng(1000 + d)
        a = rng.standard_normal((d, d)); pm = a @ a.T / d + np.eye(d)
        mu = rng.standard_normal(d) if name in ("mh_dense_d256_stream_mean", "mala_dense_d200_stream_tuned") else None
        t = K.GaussDenseTarget(pm, const=-1.5, mu=mu)
        n = 35
        x0 = rng.standard_normal((n, d)) + (0.0 if mu is None else mu[None, :])
        kw = {"mala_dense_d200_stream_tuned": dict(sampler=L.SAMPLER_MALA, nsteps=40, burnin=20, driftstep=0.05, tuner=L.TUNER_ACCEPT_RATE, targetrate=0.574, period=5),
              "mh_dense_d256_stream_mean": dict(sampler=L.SAMPLER_MH, nsteps=30, burnin=4, thinning=2, mh_sigma=np.linspace(0.02, 0.08, 256)),
              "mala_dense_d130_stream_pooled": dict(sampler=L.SAMPLER_MALA, nsteps=45, burnin=30, driftstep=0.2, tuner=L.TUNER_ACCEPT_RATE, tuner_mode=L.TUNE_POOLED,
                                                    targetrate=0.574, period=10),
              "hmc_dense_d130_dualavg_wide": dict(sampler=L.SAMPLER_HMC, nsteps=14, burnin=0, leapstep=0.1, nleaps=3, tuner=L.TUNER_DUAL_AVERAGING, targetrate=0.8,
                                                  da_nadapt=8)}[name]
        c = dict(target=t, nchains=n, x0=x0, **kw)
    elif name in SPLIT_CASES:
        # round 6: the dense Gaussian beyond D = 256 — a workgroup of W = ceil(D / 64) wavefronts per tile of 16 chains (klara_dense_split.h, layout kind 6)
        d, kw = SPLIT_CASES[name][0], dict(SPLIT_CASES[name][1])
        rng = np.random.default_rng(2000 + d)
        # (a precision matrix built from element-wise operations only: a matrix product takes whatever summation order the box's BLAS has, and the golden
        # fixtures of these cases must not depend on the box — diagonal + three dense rank-one terms: every entry non-zero, positive definite)
        pm = np.diag(1.0 + rng.random(d))
        for _ in range(3):
            u = rng.standard_normal(d)
            pm = pm + np.outer(u, u) * (0.4 / d)
        mu = rng.standard_normal(d) if "_mean" in name else None
        t = K.GaussDenseTarget(pm, const=0.75, mu=mu)
        n = kw.pop("nchains", 21)
        x0 = rng.standard_normal((n, d)) + (0.0 if mu is None else mu[None, :])
        if kw.get("sampler") == L.SAMPLER_MH:
            kw["mh_sigma"] = np.linspace(0.01, 0.04, d)
        if kw.get("sampler") == L.SAMPLER_SLICE:
            kw["slice_widths"] = np.linspace(0.5, 2.0, d)
        c = dict(target=t, nchains=n, x0=x0, **kw)
    elif name == "mh_dense_d130_wide":     # (round 4: MH at D = 130 runs on the streamed matrix-core layout too; the closure form: hmc_dense_d130_dualavg_wide)
        rng = np.random.default_rng(130)
        a = rng.standard_normal((130, 130)); p = a @ a.T / 130 + np.eye(130)
        t = K.GaussDenseTarget(p, const=0.5, mu=rng.standard_normal(130))
        c = dict(sampler=L.SAMPLER_MH, target=t, nchains=9, nsteps=12, burnin=2, mh_sigma=np.full(130, 0.08), x0=t.mu[None, :] + rng.standard_normal((9, 130)))
    elif name == "mala_dense_d100":
        c = dict(sampler=L.SAMPLER_MALA, target=K.GaussDenseTarget(compound_symmetric_precision(100)), nchains=35,
                 nsteps=20, burnin=0, driftstep=0.3)
    elif name == "mh_dense_d20":
        c = dict(sampler=L.SAMPLER_MH, target=K.GaussDenseTarget(compound_symmetric_precision(20, 0.3)), nchains=21,
                 nsteps=40, burnin=0, mh_sigma=np.full(20, 0.3))
    elif name == "mala_swiss":         # BASELINE cfg 4 shape at parity-test size
        X, y = swiss_data()
        x0 = np.array([5.1, -0.9, 8.2, -4.5])
        c = dict(sampler=L.SAMPLER_MALA, target=K.LogisticTarget(X, y, 100.0), nchains=70, nsteps=40, burnin=10,
                 driftstep=0.1, x0=x0[None, :] + 0.1 * np.random.default_rng(1).standard_normal((70, 4)))
    elif name in ("mala_logit_d2", "hmc_logit_d7", "mh_logit_d8_small", "mala_logit_d6_bigdata", "slice_logit_d3", "mala_logit_d12_wide",
                  "hmc_logit_d20_wide", "hmc_logit_d16_rows", "mh_logit_d9_rows", "slice_logit_d13_rows", "mala_logit_d11_unsplit", "mala_logit_d12_manyrows", "mala_logit_d300_closure"):
        # synthetic logistic data: E = 2 / 4 / 8 (D = 3, 6, 7: rows padded to E columns in LDS), with and without row split; 1,500 x 9
        # doubles = 108 KB of rows: beyond the 56 KB a launch gets by default
        d, nd = {"mala_logit_d2": (2, 90), "hmc_logit_d7": (7, 131), "mh_logit_d8_small": (8, 30), "mala_logit_d6_bigdata": (6, 1500),
                 "slice_logit_d3": (3, 75), "mala_logit_d12_wide": (12, 150), "hmc_logit_d20_wide": (20, 400),   # D > 16: the closure form in rounds 1-5, the matrix cores since round 6
                 # round 4: 9 .. 16 parameters on the row-split kernels (E = 16: two Philox blocks per lane, accept slots 5 .. 8), unsplit below 64
                 # rows, and back on the closure form when the rows do not fit the LDS
                 "hmc_logit_d16_rows": (16, 150), "mh_logit_d9_rows": (9, 131), "slice_logit_d13_rows": (13, 70), "mala_logit_d11_unsplit": (11, 40),
                 "mala_logit_d12_manyrows": (12, 1100),
                 # round 6: beyond the matrix-core layout's 256 parameters the closure form (one chain per lane, the vector in scratch), to 1,024 since whole-vector closures go there
                 "mala_logit_d300_closure": (300, 40)}[name]
        rng = np.random.default_rng(d)
        X = rng.standard_normal((nd, d)); beta = rng.standard_normal(d)
        y = (rng.random(nd) < 1.0 / (1.0 + np.exp(-X @ beta))).astype(np.float64)
        kw = {"mala_logit_d2": dict(sampler=L.SAMPLER_MALA, driftstep=0.05), "hmc_logit_d7": dict(sampler=L.SAMPLER_HMC, leapstep=0.05, nleaps=4),
              "mh_logit_d8_small": dict(sampler=L.SAMPLER_MH, mh_sigma=np.full(8, 0.2)),
              "mala_logit_d6_bigdata": dict(sampler=L.SAMPLER_MALA, driftstep=0.002),
              "slice_logit_d3": dict(sampler=L.SAMPLER_SLICE, slice_widths=np.full(3, 0.8)),
              "mala_logit_d12_wide": dict(sampler=L.SAMPLER_MALA, driftstep=0.01, tuner=L.TUNER_ACCEPT_RATE, targetrate=0.574, period=5),
              "hmc_logit_d20_wide": dict(sampler=L.SAMPLER_HMC, leapstep=0.07, nleaps=5),
              "hmc_logit_d16_rows": dict(sampler=L.SAMPLER_HMC, leapstep=0.05, nleaps=4, tuner=L.TUNER_DUAL_AVERAGING, targetrate=0.65, da_nadapt=15),
              "mh_logit_d9_rows": dict(sampler=L.SAMPLER_MH, mh_sigma=np.full(9, 0.15)),
              "slice_logit_d13_rows": dict(sampler=L.SAMPLER_SLICE, slice_widths=np.full(13, 0.7)),
              "mala_logit_d11_unsplit": dict(sampler=L.SAMPLER_MALA, driftstep=0.02),
              "mala_logit_d12_manyrows": dict(sampler=L.SAMPLER_MALA, driftstep=0.002),
              "mala_logit_d300_closure": dict(sampler=L.SAMPLER_MALA, driftstep=0.002)}[name]
        c = dict(target=K.LogisticTarget(X, y, 10.0), nchains=45, nsteps=25, burnin=5, x0=0.1 * rng.standard_normal((45, d)), **kw)
    elif name in LOGIT_MFMA_CASES:
        # round 6: the logistic regression beyond 16 parameters on the matrix cores (klara_logit_mfma.h, layout kind 5): D = 17 .. 128 (NE = 8, 16, 24, 32
        # elements per lane), data rows that do not fill their last tile of 16 / their last block of tiles, every sampler it serves, tuners, monitors
        d, nd, kw = LOGIT_MFMA_CASES[name]
        rng = np.random.default_rng(1000 + d + nd)
        X = rng.standard_normal((nd, d)) / np.sqrt(d); beta = rng.standard_normal(d)
        y = (rng.random(nd) < 1.0 / (1.0 + np.exp(-X @ beta))).astype(np.float64)
        kw = dict(kw)
        if kw.get("mh_sigma") is not None:
            kw["mh_sigma"] = np.full(d, kw["mh_sigma"])
        if kw.get("slice_widths") is not None:
            kw["slice_widths"] = np.full(d, kw["slice_widths"])
        nch = kw.pop("nchains", 45)
        c = dict(target=K.LogisticTarget(X, y, 10.0), nchains=nch, nsteps=kw.pop("nsteps", 25), burnin=kw.pop("burnin", 5), x0=0.1 * rng.standard_normal((nch, d)), **kw)
    elif name == "hmc_swiss":
        X, y = swiss_data()
        c = dict(sampler=L.SAMPLER_HMC, target=K.LogisticTarget(X, y, 100.0), nchains=65, nsteps=12, burnin=0,
                 leapstep=0.05, nleaps=6, x0=0.1 * np.random.default_rng(2).standard_normal((65, 4)))
    elif name == "hmc_rats":          # BASELINE cfg 5 shape at parity-test size (L reduced from 32 to 8)
        t = rats_target()
        x0 = t.least_squares_start()[None, :] + 0.05 * np.random.default_rng(4).standard_normal((37, t.ndims))
        c = dict(sampler=L.SAMPLER_HMC, target=t, nchains=37, nsteps=30, burnin=5, leapstep=0.01, nleaps=8, x0=x0)
    elif name == "hmc_rats_pooled":   # per-GPU pooled AcceptanceRateMCTuner, as cfg 5 asks
        t = rats_target()
        x0 = t.least_squares_start()[None, :] + 0.05 * np.random.default_rng(5).standard_normal((70, t.ndims))
        c = dict(sampler=L.SAMPLER_HMC, target=t, nchains=70, nsteps=90, burnin=60, leapstep=0.02, nleaps=6,
                 tuner=L.TUNER_ACCEPT_RATE, tuner_mode=L.TUNE_POOLED, targetrate=0.65, period=20, x0=x0)
    elif name == "mala_rats":
        t = rats_target()
        x0 = t.least_squares_start()[None, :] + 0.05 * np.random.default_rng(6).standard_normal((33, t.ndims))
        c = dict(sampler=L.SAMPLER_MALA, target=t, nchains=33, nsteps=40, burnin=0, driftstep=1e-4, x0=x0)
    elif name == "slice_rats":
        t = rats_target()
        x0 = t.least_squares_start()[None, :] + 0.05 * np.random.default_rng(7).standard_normal((6, t.ndims))
        c = dict(sampler=L.SAMPLER_SLICE, target=t, nchains=6, nsteps=3, burnin=0, slice_widths=np.full(t.ndims, 0.5), x0=x0)
    elif name == "slice_d5":           # test/SliceSampler.jl-style: SliceSampler(1., 5)
        c = dict(sampler=L.SAMPLER_SLICE, target=K.GaussDiagTarget.negdot(5), nchains=40, nsteps=30, burnin=5,
                 slice_widths=np.full(5, 1.0), slice_stepout=True)
    elif name == "slice_d100_nostepout":
        c = dict(sampler=L.SAMPLER_SLICE, target=K.GaussDiagTarget.negdot(100), nchains=6, nsteps=4, burnin=0,
                 slice_widths=np.full(100, 3.0), slice_stepout=False)
    elif name == "mh_rats":            # MH on the hierarchical model (per-element proposal scales), few-lanes layout
        t = rats_target()
        x0 = t.least_squares_start()[None, :] + 0.05 * np.random.default_rng(9).standard_normal((29, t.ndims))
        c = dict(sampler=L.SAMPLER_MH, target=t, nchains=29, nsteps=40, burnin=5, mh_sigma=np.linspace(0.02, 0.2, t.ndims), x0=x0)
    elif name == "mala_rats_tuned":    # AcceptanceRate per chain on the few-lanes layout
        t = rats_target()
        x0 = t.least_squares_start()[None, :] + 0.05 * np.random.default_rng(10).standard_normal((21, t.ndims))
        c = dict(sampler=L.SAMPLER_MALA, target=t, nchains=21, nsteps=60, burnin=40, driftstep=2e-3, tuner=L.TUNER_ACCEPT_RATE,
                 targetrate=0.574, period=10, x0=x0)
    elif name in ("slice_dense_d20", "slice_dense_d37_mean", "slice_dense_d192_stream", "slice_dense_d130_stream_mean", "slice_dense_d256_stream"):
        # slice sampler on the dense target: every probe is a full MFMA evaluation (round 5: beyond D = 128 on the streamed layouts, the chains of a tile out of lockstep)
        d = int(name.split("_")[2][1:])
        rng = np.random.default_rng(300 + d)
        a = rng.standard_normal((d, d)); p = a @ a.T / d + np.eye(d)
        mean = name.endswith("mean")
        t = K.GaussDenseTarget(p, const=-1.25, mu=rng.standard_normal(d) if mean else None)
        c = dict(sampler=L.SAMPLER_SLICE, target=t, nchains=21, nsteps=6 if d <= 128 else 3, burnin=1, slice_widths=np.linspace(0.4, 2.5, d),
                 slice_stepout=not mean, x0=rng.standard_normal((21, d)) + (t.mu if mean else 0.0))
    elif name == "slice_d20_stepout":  # pair-transposed layout: step-out and shrink loops with per-chain trip counts, non-unit diagonal
        c = dict(sampler=L.SAMPLER_SLICE, target=K.GaussDiagTarget.mvnormal(np.linspace(-1, 2, 20), np.linspace(0.5, 3.0, 20)), nchains=37,
                 nsteps=8, burnin=2, slice_widths=np.linspace(0.2, 4.0, 20))
    elif name == "slice_swiss":        # doc/examples/swiss/SliceSampler.jl shape
        X, y = swiss_data()
        c = dict(sampler=L.SAMPLER_SLICE, target=K.LogisticTarget(X, y, 100.0), nchains=70, nsteps=6, burnin=0,
                 slice_widths=np.full(4, 1.0), x0=0.1 * np.random.default_rng(3).standard_normal((70, 4)))
    # ---- pair-transposed layout (kind 3): VanillaMCTuner jobs on diagonal Gaussians that monitor at most the accept mask
    elif name == "dt_mala_d100":       # BASELINE cfg 2 shape; 130 chains = 16 full wavefront groups + 2 chains
        c = dict(sampler=L.SAMPLER_MALA, target=K.GaussDiagTarget.negdot(100), nchains=130, nsteps=40, burnin=0, driftstep=0.9)
    elif name in ("sparse_mala_d100", "sparse_mala_d100_small_step", "sparse_mh_d100", "sparse_mala_mvnormal_d30"):
        # klara_desc.sparse_moves = 1: untuned MH / MALA that keep running sums always on the 4-lane kernels, sums folded by atomic adds —
        # at a low acceptance (the layout's purpose), at a high one (a fold at almost every transition) and on a non-unit diagonal
        base = {"sparse_mala_d100": dict(sampler=L.SAMPLER_MALA, target=K.GaussDiagTarget.negdot(100), nchains=131, nsteps=60, burnin=10, driftstep=0.9),
                "sparse_mala_d100_small_step": dict(sampler=L.SAMPLER_MALA, target=K.GaussDiagTarget.negdot(100), nchains=67, nsteps=40, burnin=0, thinning=1, driftstep=0.05),
                "sparse_mh_d100": dict(sampler=L.SAMPLER_MH, target=K.GaussDiagTarget.negdot(100), nchains=70, nsteps=60, burnin=7, thinning=3, mh_sigma=np.full(100, 0.1)),
                "sparse_mala_mvnormal_d30": dict(sampler=L.SAMPLER_MALA, target=K.GaussDiagTarget.mvnormal(np.linspace(-1, 2, 30), np.linspace(0.6, 1.7, 30)),
                                                 nchains=45, nsteps=50, burnin=5, driftstep=0.3)}[name]
        c = dict(base, sparse_moves=1)
    elif name == "dt_mala_d100_small_step":
        c = dict(sampler=L.SAMPLER_MALA, target=K.GaussDiagTarget.negdot(100), nchains=67, nsteps=40, burnin=0, driftstep=0.05)
    elif name == "dt_mala_d112_full":  # D/2 = 7*8: no padding pair, the accept uniform takes the explicit path
        c = dict(sampler=L.SAMPLER_MALA, target=K.GaussDiagTarget.negdot(112), nchains=19, nsteps=30, burnin=0, driftstep=0.04)
    elif name == "dt_mala_mvnormal_d30":
        mu = np.linspace(-1, 2, 30); sg = np.linspace(0.6, 1.7, 30)
        c = dict(sampler=L.SAMPLER_MALA, target=K.GaussDiagTarget.mvnormal(mu, sg), nchains=45, nsteps=50, burnin=0, driftstep=0.3)
    elif name == "dt_mala_d18":        # smallest dimension on this layout (below it the group layout packs 16..64 chains per wavefront)
        c = dict(sampler=L.SAMPLER_MALA, target=K.GaussDiagTarget.negdot(18), nchains=100, nsteps=60, burnin=0, driftstep=0.5)
    elif name == "dt_mh_d100":
        c = dict(sampler=L.SAMPLER_MH, target=K.GaussDiagTarget.negdot(100), nchains=70, nsteps=60, burnin=0,
                 mh_sigma=np.full(100, 0.1))
    elif name == "dt_mh_mvnormal_d20":
        mu = np.linspace(-2, 3, 20); sg = np.linspace(0.5, 2.0, 20)
        c = dict(sampler=L.SAMPLER_MH, target=K.GaussDiagTarget.mvnormal(mu, sg), nchains=33, nsteps=80, burnin=0,
                 mh_sigma=sg * 0.4)
    elif name == "dt_hmc_d100":
        c = dict(sampler=L.SAMPLER_HMC, target=K.GaussDiagTarget.negdot(100), nchains=70, nsteps=20, burnin=0,
                 leapstep=0.1, nleaps=10)
    elif name == "dt_hmc_d128_full":   # D/2 = 8*8: no padding pair
        c = dict(sampler=L.SAMPLER_HMC, target=K.GaussDiagTarget.negdot(128), nchains=21, nsteps=15, burnin=0,
                 leapstep=0.3, nleaps=5)
    elif name == "dt_hmc_mvnormal_d96":
        mu = np.linspace(-1, 1, 96); sg = np.linspace(0.7, 1.4, 96)
        c = dict(sampler=L.SAMPLER_HMC, target=K.GaussDiagTarget.mvnormal(mu, sg), nchains=29, nsteps=15, burnin=0,
                 leapstep=0.15, nleaps=7)
    elif name == "custom_negdot_mala_d3":     # same arithmetic as the built-in diagonal target at D = 3 (E = 4, one chain per lane)
        c = dict(sampler=L.SAMPLER_MALA, target=K.CustomTarget(3, SRC_NEGDOT), nchains=70, nsteps=60, burnin=10, driftstep=0.8)
    elif name == "custom_banana_mh":          # logtarget closure only
        c = dict(sampler=L.SAMPLER_MH, target=K.CustomTarget(2, SRC_BANANA_LT_ONLY), nchains=130, nsteps=80, burnin=20,
                 mh_sigma=[0.7, 0.9], x0=np.tile([0.5, 0.2], (130, 1)))
    elif name == "custom_banana_hmc":
        c = dict(sampler=L.SAMPLER_HMC, target=K.CustomTarget(2, SRC_BANANA), nchains=67, nsteps=60, burnin=10, leapstep=0.15, nleaps=7,
                 x0=np.tile([0.5, 0.2], (67, 1)))
    elif name == "custom_banana_slice":
        c = dict(sampler=L.SAMPLER_SLICE, target=K.CustomTarget(2, SRC_BANANA_LT_ONLY), nchains=65, nsteps=30, burnin=5,
                 slice_widths=[1.0, 2.0], x0=np.tile([0.5, 0.2], (65, 1)))
    elif name == "custom_logit_mala_d4":      # the swiss example's closures on 40 synthetic rows (= built-in LogisticTarget without row split)
        X, y = synthetic_logit(40, 4)
        c = dict(sampler=L.SAMPLER_MALA, target=K.CustomTarget(4, SRC_LOGIT, np.concatenate([X.ravel(), y, [100.0]])),
                 nchains=64, nsteps=50, burnin=10, driftstep=0.05, x0=np.zeros((64, 4)))
    elif name == "custom_quartic_hmc_d10_dualavg":
        c = dict(sampler=L.SAMPLER_HMC, target=K.CustomTarget(10, SRC_QUARTIC_CHAIN, [0.1, 0.7]), nchains=40, nsteps=70, burnin=0,
                 leapstep=0.1, nleaps=6, tuner=L.TUNER_DUAL_AVERAGING, targetrate=0.65, da_nadapt=40)
    elif name == "custom_quartic_mala_d20_pooled":
        c = dict(sampler=L.SAMPLER_MALA, target=K.CustomTarget(20, SRC_QUARTIC_CHAIN, [0.05, 0.3]), nchains=48, nsteps=130, burnin=100,
                 driftstep=0.2, tuner=L.TUNER_ACCEPT_RATE, tuner_mode=L.TUNE_POOLED, targetrate=0.574, period=20)
    elif name == "custom_quartic_hmc_d32":    # 32 elements per lane: still register-resident
        c = dict(sampler=L.SAMPLER_HMC, target=K.CustomTarget(32, SRC_QUARTIC_CHAIN, [0.02, 0.5]), nchains=70, nsteps=25, burnin=5,
                 leapstep=0.1, nleaps=4)
    elif name == "custom_quartic_mala_d64":   # beyond 32 dimensions: 4 lanes per chain, staged through LDS
        c = dict(sampler=L.SAMPLER_MALA, target=K.CustomTarget(64, SRC_QUARTIC_CHAIN, [0.02, 0.5]), nchains=70, nsteps=20, burnin=5,
                 driftstep=0.05)
    elif name == "custom_quartic_mala_d100":  # a user-defined target at the BASELINE dimension: 8 lanes per chain, staged through LDS
        c = dict(sampler=L.SAMPLER_MALA, target=K.CustomTarget(100, SRC_QUARTIC_CHAIN, [0.02, 0.5]), nchains=70, nsteps=12, burnin=2,
                 driftstep=0.04)
    # ---- round 6: whole-vector closures beyond 256 dimensions (refused before): the staged form on 32 / 64 lanes per chain, 16 elements per lane
    elif name == "custom_quartic_hmc_d300":   # 32 lanes per chain
        c = dict(sampler=L.SAMPLER_HMC, target=K.CustomTarget(300, SRC_QUARTIC_CHAIN, [0.02, 0.5]), nchains=9, nsteps=10, burnin=2,
                 leapstep=0.06, nleaps=3, tuner=L.TUNER_ACCEPT_RATE, targetrate=0.7, period=4)
    elif name == "custom_quartic_mala_d700":  # 64 lanes per chain: one chain per wavefront
        c = dict(sampler=L.SAMPLER_MALA, target=K.CustomTarget(700, SRC_QUARTIC_CHAIN, [0.02, 0.5]), nchains=7, nsteps=10, burnin=2, thinning=2,
                 driftstep=0.01)
    elif name == "custom_negdot_mh_d1024":    # the largest
        c = dict(sampler=L.SAMPLER_MH, target=K.CustomTarget(1024, SRC_NEGDOT), nchains=5, nsteps=12, burnin=0, mh_sigma=np.full(1024, 0.03))
    elif name == "custom_quartic_hmc_d50":
        c = dict(sampler=L.SAMPLER_HMC, target=K.CustomTarget(50, SRC_QUARTIC_CHAIN, [0.02, 0.5]), nchains=33, nsteps=12, burnin=2,
                 leapstep=0.08, nleaps=5)
    elif name == "custom_normal_normal_mala":   # likelihood + prior closures (lt = ll + lp, grad = gll + glp composed by the library)
        rng = np.random.default_rng(12)
        t = normal_normal_target(rng.standard_normal(6) * 2, np.linspace(0.5, 2.0, 6), np.linspace(-1, 1, 6), np.linspace(1.0, 5.0, 6))
        c = dict(sampler=L.SAMPLER_MALA, target=t, nchains=70, nsteps=60, burnin=10, driftstep=0.4, x0=np.zeros((70, 6)))
    elif name == "custom_normal_normal_mh":     # no gradient closures needed
        t = K.CustomTarget.likelihood_prior(2, SRC_NN_LL, SRC_NN_LP, data=np.array([-1.88, 2.23, 1.0, 1.0, 0.0, 0.0, 1.0, 1.0]))
        c = dict(sampler=L.SAMPLER_MH, target=t, nchains=66, nsteps=50, burnin=10, mh_sigma=[0.8, 0.8], x0=np.tile([-2.637, -1.132], (66, 1)))
    # ---- whole-vector closures beyond 32 dimensions: G lanes per chain, evaluations staged through LDS (klara_custom.h STAGED)
    elif name == "staged_negdot_mala_d100_big_step":   # the README closure at BASELINE cfg 2's shape and drift step (rare accepts)
        c = dict(sampler=L.SAMPLER_MALA, target=K.CustomTarget(100, SRC_NEGDOT), nchains=131, nsteps=60, burnin=0, driftstep=0.9)
    elif name == "staged_quartic_mh_d33_thinned":      # odd D (the last lane's elements end early), MH needs no gradient closure
        c = dict(sampler=L.SAMPLER_MH, target=K.CustomTarget(33, SRC_QUARTIC_CHAIN[:SRC_QUARTIC_CHAIN.index("KLARA_USER_FN void")], [0.1, 0.4]), nchains=45, nsteps=50,
                 burnin=4, thinning=2, mh_sigma=np.linspace(0.05, 0.4, 33), x0=0.3 * np.random.default_rng(5).standard_normal((45, 33)))
    elif name == "staged_quartic_slice_d40":
        c = dict(sampler=L.SAMPLER_SLICE, target=K.CustomTarget(40, SRC_QUARTIC_CHAIN[:SRC_QUARTIC_CHAIN.index("KLARA_USER_FN void")], [0.05, 0.3]), nchains=19, nsteps=6,
                 burnin=1, slice_widths=np.linspace(0.6, 1.4, 40))
    elif name == "staged_quartic_hmc_d200_dualavg":    # 16 lanes per chain
        c = dict(sampler=L.SAMPLER_HMC, target=K.CustomTarget(200, SRC_QUARTIC_CHAIN, [0.02, 0.5]), nchains=21, nsteps=50, burnin=0,
                 leapstep=0.05, nleaps=5, tuner=L.TUNER_DUAL_AVERAGING, targetrate=0.65, da_nadapt=30)
    elif name == "staged_quartic_mala_d70_pooled":
        c = dict(sampler=L.SAMPLER_MALA, target=K.CustomTarget(70, SRC_QUARTIC_CHAIN, [0.05, 0.3]), nchains=48, nsteps=130, burnin=100,
                 driftstep=0.1, tuner=L.TUNER_ACCEPT_RATE, tuner_mode=L.TUNE_POOLED, targetrate=0.574, period=20)
    elif name == "staged_normal_normal_mala_d48":      # likelihood + prior closures: three rows of LDS per chain
        rng = np.random.default_rng(13)
        t = normal_normal_target(rng.standard_normal(48) * 2, np.linspace(0.5, 2.0, 48), np.linspace(-1, 1, 48), np.linspace(1.0, 5.0, 48))
        c = dict(sampler=L.SAMPLER_MALA, target=t, nchains=70, nsteps=40, burnin=10, driftstep=0.1, x0=np.zeros((70, 48)))
    elif name == "staged_quartic_hmc_d256_tuned":      # the largest user-defined dimension: 32 lanes per chain
        c = dict(sampler=L.SAMPLER_HMC, target=K.CustomTarget(256, SRC_QUARTIC_CHAIN, [0.02, 0.5]), nchains=9, nsteps=60, burnin=40,
                 leapstep=0.05, nleaps=3, tuner=L.TUNER_ACCEPT_RATE, targetrate=0.7, period=10)
    # ---- pair closures on the pair-transposed layout
    elif name == "pair_negdot_mala_d100":      # the README closure, BASELINE cfg 2's shape, ragged chain count
        c = dict(sampler=L.SAMPLER_MALA, target=K.CustomTarget.pairwise(100, SRC_PAIR_NEGDOT), nchains=67, nsteps=40, burnin=5, driftstep=0.05)
    elif name == "pair_negdot_mala_d100_big_step":     # ... at the drift step of cfg 2 (rare accepts), one transition per launch exercised by the modes
        c = dict(sampler=L.SAMPLER_MALA, target=K.CustomTarget.pairwise(100, SRC_PAIR_NEGDOT), nchains=130, nsteps=60, burnin=0, driftstep=0.9)
    elif name == "pair_quartic_hmc_d50_tuned":
        c = dict(sampler=L.SAMPLER_HMC, target=K.CustomTarget.pairwise(50, SRC_PAIR_QUARTIC, [0.02, 0.5]), nchains=33, nsteps=130, burnin=100,
                 leapstep=0.15, nleaps=4, tuner=L.TUNER_ACCEPT_RATE, targetrate=0.7, period=25, x0=0.5 * np.random.default_rng(4).standard_normal((33, 50)))
    elif name == "pair_banana_mh_d33":         # odd D: the last pair is half a pair
        c = dict(sampler=L.SAMPLER_MH, target=K.CustomTarget.pairwise(33, SRC_PAIR_QUARTIC, [0.1, 0.4]), nchains=45, nsteps=50, burnin=4, thinning=2,
                 mh_sigma=np.linspace(0.05, 0.4, 33), x0=0.3 * np.random.default_rng(5).standard_normal((45, 33)))
    elif name == "pair_banana_hmc_d100_dualavg":
        c = dict(sampler=L.SAMPLER_HMC, target=K.CustomTarget.pairwise(100, SRC_PAIR_BANANA, [0.05, 9.0]), nchains=21, nsteps=60, burnin=0,
                 leapstep=0.05, nleaps=6, tuner=L.TUNER_DUAL_AVERAGING, targetrate=0.65, da_nadapt=40,
                 x0=np.random.default_rng(6).standard_normal((21, 100)) * np.tile([2.0, 1.0], 50))
    elif name == "pair_quartic_mala_d300_pooled":      # 32 lanes per chain, pooled tuner
        c = dict(sampler=L.SAMPLER_MALA, target=K.CustomTarget.pairwise(300, SRC_PAIR_QUARTIC, [0.05, 0.3]), nchains=19, nsteps=120, burnin=100,
                 driftstep=0.02, tuner=L.TUNER_ACCEPT_RATE, tuner_mode=L.TUNE_POOLED, targetrate=0.574, period=20,
                 x0=0.4 * np.random.default_rng(7).standard_normal((19, 300)))
    elif name == "pair_indexed_mala_d100":     # data indexed by coordinate: D = 100 on 8 lanes has padding pairs 50..55
        c = dict(sampler=L.SAMPLER_MALA, target=K.CustomTarget.pairwise(100, SRC_PAIR_INDEXED, np.linspace(0.5, 2.0, 100)), nchains=67, nsteps=40, burnin=5,
                 driftstep=0.05)
    elif name == "pair_indexed_hmc_d37":       # odd D: a half pair, and padding pairs 19..23
        c = dict(sampler=L.SAMPLER_HMC, target=K.CustomTarget.pairwise(37, SRC_PAIR_INDEXED, np.linspace(0.5, 2.0, 37)), nchains=35, nsteps=30, burnin=5,
                 leapstep=0.1, nleaps=5)
    # ---- pair closures the pair-transposed kernels do not serve (fewer than 9 pairs; the slice sampler): taken as whole-vector closures (round 5; were refused)
    elif name == "pair_quartic_mala_d9_whole":         # odd D below 17: one chain per lane
        c = dict(sampler=L.SAMPLER_MALA, target=K.CustomTarget.pairwise(9, SRC_PAIR_QUARTIC, [0.1, 0.4]), nchains=70, nsteps=40, burnin=5, driftstep=0.05,
                 x0=0.3 * np.random.default_rng(11).standard_normal((70, 9)))
    elif name == "pair_banana_hmc_d16_whole":
        c = dict(sampler=L.SAMPLER_HMC, target=K.CustomTarget.pairwise(16, SRC_PAIR_BANANA, [0.05, 9.0]), nchains=33, nsteps=30, burnin=0, leapstep=0.05, nleaps=5,
                 tuner=L.TUNER_ACCEPT_RATE, targetrate=0.7, period=10, x0=np.random.default_rng(12).standard_normal((33, 16)) * np.tile([2.0, 1.0], 8))
    elif name == "pair_indexed_slice_d40_whole":       # the slice sampler on a pair closure (round 5: staged whole-vector form on 4 lanes per chain; round 6: the few-lanes kernels, the whole-vector form under KLARA_PAIR_SLICE_AS_WHOLE=1)
        c = dict(sampler=L.SAMPLER_SLICE, target=K.CustomTarget.pairwise(40, SRC_PAIR_INDEXED, np.linspace(0.5, 2.0, 40)), nchains=21, nsteps=6, burnin=1,
                 slice_widths=np.linspace(0.5, 2.5, 40))
    elif name == "pair_negdot_slice_d6_whole":
        c = dict(sampler=L.SAMPLER_SLICE, target=K.CustomTarget.pairwise(6, SRC_PAIR_NEGDOT), nchains=66, nsteps=10, burnin=2, slice_widths=np.full(6, 1.5), slice_stepout=False)
    # ---- round 6: the slice sampler on a pair closure runs on the few-lanes kernels (k_diagt<SLICE, .., USERPAIR>: a probe compares the pair's own term)
    elif name == "pair_quartic_slice_d100":            # coupled within the pair: coordinate 2P + 1 sees the new x_2P; step-out on
        c = dict(sampler=L.SAMPLER_SLICE, target=K.CustomTarget.pairwise(100, SRC_PAIR_QUARTIC, [0.1, 0.4]), nchains=37, nsteps=8, burnin=2,
                 slice_widths=np.linspace(0.5, 2.5, 100), x0=0.5 * np.random.default_rng(21).standard_normal((37, 100)))
    elif name == "pair_banana_slice_d37":              # odd D: a half pair (x1 = 0) and padding pairs; no step-out; thinning
        c = dict(sampler=L.SAMPLER_SLICE, target=K.CustomTarget.pairwise(37, SRC_PAIR_BANANA, [0.05, 9.0]), nchains=21, nsteps=12, burnin=3, thinning=2,
                 slice_widths=np.full(37, 2.0), slice_stepout=False, x0=np.random.default_rng(22).standard_normal((21, 37)))
    elif name == "pair_indexed_slice_d300":            # 32 lanes per chain
        c = dict(sampler=L.SAMPLER_SLICE, target=K.CustomTarget.pairwise(300, SRC_PAIR_INDEXED, np.linspace(0.5, 2.0, 300)), nchains=9, nsteps=5, burnin=1,
                 slice_widths=np.linspace(0.5, 2.5, 300))
    elif name == "custom_quartic_slice_d7":
        c = dict(sampler=L.SAMPLER_SLICE, target=K.CustomTarget(7, SRC_QUARTIC_CHAIN, [0.1, 0.4]), nchains=66, nsteps=12, burnin=2,
                 slice_widths=np.full(7, 1.5))
    else:
        raise KeyError(name)
    c.setdefault("x0", None)
    c.setdefault("seed", 20260927)
    c["name"] = name
    return c


DIAGT_CASES = ["dt_mala_d100", "dt_mala_d100_small_step", "dt_mala_d112_full", "dt_mala_mvnormal_d30", "dt_mala_d18",
               "dt_mh_d100", "dt_mh_mvnormal_d20", "dt_hmc_d100", "dt_hmc_d128_full", "dt_hmc_mvnormal_d96"]

ALL_CASES = ["mh_readme", "mh_d100", "mh_mvnormal_d7", "mala_d100", "mala_d100_small_step", "mala_d3_tuned",
             "mala_d300", "hmc_d100", "hmc_d10_tuned_pooled", "hmc_dense_d100", "hmc_dense_d37", "mala_dense_d100",
             "mh_dense_d20", "mala_swiss", "hmc_swiss", "slice_d5", "slice_d100_nostepout", "slice_swiss",
             "hmc_rats", "hmc_rats_pooled", "mala_rats", "slice_rats", "hmc_d10_dualavg", "hmc_dense_d37_dualavg",
             "hmc_rats_dualavg", "mala_d3_tuned_erf", "mala_d1", "hmc_d128_full", "mala_d129", "mh_d512",
             "hmc_d1000_tuned", "mala_d777_pooled", "slice_d600", "pair_quartic_hmc_d700_dualavg",
             "slice_d2_mvnormal", "mala_d20_tuned", "hmc_d100_tuned", "mala_d100_verbose", "hmc_d40_dualavg", "hmc_d100_dualavg",
             "hmc_dense_d98", "hmc_dense_d70", "hmc_dense_d128", "hmc_dense_d100_mean", "mala_dense_d37_mean", "mh_dense_d20_mean",
             "hmc_dense_d70_mean_dualavg", "slice_dense_d20", "slice_dense_d37_mean", "mh_dense_d130_wide", "hmc_dense_d130_wide",
             "hmc_dense_d256_stream_tuned", "hmc_dense_d192_stream_mean", "hmc_dense_d160_stream_pooled", "hmc_dense_d129_stream", "hmc_dense_d201_stream",
             "mala_dense_d200_stream_tuned", "mh_dense_d256_stream_mean", "mala_dense_d130_stream_pooled", "hmc_dense_d130_dualavg_wide",
             "hmc_dense_d300_split", "hmc_dense_d257_split_pooled", "mala_dense_d512_split_mean_tuned", "mh_dense_d700_split_mean",
             "hmc_dense_d1024_split_dualavg", "mala_dense_d1000_split_pooled", "mh_dense_d333_split", "slice_dense_d260_split", "slice_dense_d530_split_mean",
             "slice_dense_d192_stream", "slice_dense_d130_stream_mean", "slice_dense_d256_stream", "mala_logit_d2", "hmc_logit_d7", "mh_logit_d8_small",
             "mala_logit_d6_bigdata", "slice_logit_d3", "mala_logit_d12_wide", "hmc_logit_d20_wide", "slice_d20_stepout", "mh_rats", "mala_rats_tuned",
             "hmc_logit_d16_rows", "mh_logit_d9_rows", "slice_logit_d13_rows", "mala_logit_d11_unsplit", "mala_logit_d12_manyrows"] + list(LOGIT_MFMA_CASES) + [
             "custom_negdot_mala_d3", "custom_banana_mh", "custom_banana_hmc", "custom_banana_slice", "custom_logit_mala_d4",
             "custom_quartic_hmc_d10_dualavg", "custom_quartic_mala_d20_pooled", "custom_quartic_hmc_d32", "custom_quartic_slice_d7",
             "custom_quartic_mala_d64", "custom_quartic_mala_d100", "custom_quartic_hmc_d50", "mala_logit_d300_closure", "custom_quartic_hmc_d300", "custom_quartic_mala_d700", "custom_negdot_mh_d1024", "custom_normal_normal_mala", "custom_normal_normal_mh",
             "sparse_mala_d100", "sparse_mala_d100_small_step", "sparse_mh_d100", "sparse_mala_mvnormal_d30",
             "pair_negdot_mala_d100", "pair_negdot_mala_d100_big_step", "pair_quartic_hmc_d50_tuned", "pair_banana_mh_d33",
             "pair_banana_hmc_d100_dualavg", "pair_quartic_mala_d300_pooled", "pair_indexed_mala_d100", "pair_indexed_hmc_d37",
             "pair_quartic_mala_d9_whole", "pair_banana_hmc_d16_whole", "pair_indexed_slice_d40_whole", "pair_negdot_slice_d6_whole",
             "pair_quartic_slice_d100", "pair_banana_slice_d37", "pair_indexed_slice_d300",
             "staged_negdot_mala_d100_big_step", "staged_quartic_mh_d33_thinned", "staged_quartic_slice_d40", "staged_quartic_hmc_d200_dualavg",
             "staged_quartic_mala_d70_pooled", "staged_normal_normal_mala_d48", "staged_quartic_hmc_d256_tuned"]
# cases whose oracle output is also committed as a golden fixture (tests/golden/<name>.npz)
GOLDEN_CASES = ["mh_readme", "mala_d100", "hmc_d100", "hmc_dense_d100", "mala_swiss", "slice_d5",
                "mala_d3_tuned", "hmc_d10_tuned_pooled", "hmc_rats", "hmc_d10_dualavg",
                "dt_mala_d100_small_step", "dt_hmc_d100", "dt_mh_mvnormal_d20", "custom_banana_hmc",
                "custom_quartic_mala_d20_pooled", "pair_quartic_hmc_d50_tuned",
                # round 5's new paths: the slice sampler on the free-running diagonal kernel, on the dense targets (chains of a tile out of lockstep; streamed layout), a pair closure run as a whole-vector closure
                "slice_d20_stepout", "slice_dense_d20", "slice_dense_d130_stream_mean", "pair_quartic_mala_d9_whole",
                # round 6: the logistic regression beyond 16 parameters on the matrix cores (layout kind 5)
                "mala_logitm_d20", "hmc_logitm_d40_dualavg",
                # ... dense targets beyond D = 256 on the workgroup-split layout (kind 6: the even tile deal) and the slice sampler on a pair closure (the pair's own term)
                "hmc_dense_d300_split", "mala_dense_d512_split_mean_tuned", "slice_dense_d260_split", "pair_quartic_slice_d100"]


def oracle_kwargs(case, layout=None, chain_offset=0, nchains=None):
    """Translate a case into OracleJob keyword arguments."""
    t = case["target"]
    kw = {k: v for k, v in case.items() if k not in ("target", "x0", "name")}
    kw["target_kind"] = t.kind
    kw["ndims"] = t.ndims
    if isinstance(t, K.GaussDiagTarget):
        kw.update(gauss_w=t.w, gauss_mu=t.mu, gauss_const=t.const)
    elif isinstance(t, K.GaussDenseTarget):
        kw.update(gauss_prec=t.precision, gauss_const=t.const, gauss_mu=t.mu)
    elif isinstance(t, K.CustomTarget):
        kw.update(custom_src=t.source, custom_data=t.data)
    elif isinstance(t, K.HierNormalTarget):
        kw.update(hier_Y=t.Y, hier_xc=t.xc, hier_prior_prec=t.prior_prec, hier_gamma_a=t.gamma_a, hier_gamma_b=t.gamma_b)
    else:
        kw.update(logit_X=t.X, logit_y=t.y, logit_lambda=t.lam)
    kw["layout"] = layout
    kw["chain_offset"] = chain_offset
    if nchains is not None:
        kw["nchains"] = nchains
    return kw


def engine_kwargs(case, monitor=L.MON_ACCEPT | L.MON_SUMMARIES, **over):
    kw = {k: v for k, v in case.items() if k not in ("x0", "name")}
    kw["monitor"] = monitor
    kw.update(over)
    return kw
