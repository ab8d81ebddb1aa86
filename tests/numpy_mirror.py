"""A second, independent restatement of the reference's transition path — plain Python / numpy, written from the Julia sources
and SURVEY.md Appendix A, sharing NO code with oracle/klara_oracle.c or klara.jl_amd/csrc/detmath.h ("C primary;
Python/numpy mirror").  Test infrastructure only.

What is restated here, and from where:
  * the counter stream: Philox4x32-10 (Salmon et al., SC'11: multipliers 0xD2511F53 / 0xCD9E8D57, Weyl
    constants 0x9E3779B9 / 0xBB67AE85), counter = ((transition << 24) | slot, global chain id), key = seed; 52-bit uniforms
    (m + 1/2) 2^-52; Box-Muller sqrt(-2 ln u1) (cos, sin)(2 pi u2) with numpy's libm — NOT the library's table-driven functions, so
    the normals agree with the library's to a few ulp only and trajectories to ~1e-12, while accept decisions must be identical;
  * iterate!(job, MH | MALA | HMC | SliceSampler, Multivariate): src/samplers/iterate/{MH.jl:72-124, MALA.jl:78-128, HMC.jl:124-201,
    SliceSampler.jl:60-109}, leapfrog! samplers.jl:122-134, hamiltonian samplers.jl:103;
  * the tuning block iterate/MALA.jl:130-152 / HMC.jl:203-224 with tuners.jl:27-32 and AcceptanceRateMCTuner.jl:9,46;
  * the save rule BasicMCJob.jl:219-238 with BasicMCRange.jl:17-36, mean(chain) stats/mean.jl:7-11;
  * the targets: README.md:23,155 (-dot(z,z)), the MvNormal closures of test/BasicContMuvParameter.jl, the swiss logistic regression
    doc/examples/swiss/MALA/analytical.jl:11-18, and the builder-defined dense Gaussian.
"""
import math

import numpy as np

M32 = 0xFFFFFFFF


# ------------------------------------------------------------------ counter stream
def philox4x32_10(c0, c1, c2, c3, k0, k1):
    for _ in range(10):
        p0 = 0xD2511F53 * c0
        p1 = 0xCD9E8D57 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & M32, p1 & M32, ((p0 >> 32) ^ c3 ^ k1) & M32, p0 & M32
        k0 = (k0 + 0x9E3779B9) & M32
        k1 = (k1 + 0xBB67AE85) & M32
    return c0, c1, c2, c3


def stream_block(seed, chain, transition, slot):
    blk = ((transition << 24) | slot) & 0xFFFFFFFFFFFFFFFF
    return philox4x32_10(blk & M32, blk >> 32, chain & M32, (chain >> 32) & M32, seed & M32, (seed >> 32) & M32)


def u52(hi, lo):
    m = (hi << 20) | (lo >> 12)
    return (m + 0.5) * 2.0 ** -52


def u44(wa, wb):
    """44 bits: word wa and the low 12 bits of wb"""
    m = (wa << 12) | (wb & 0xFFF)
    return (m + 0.5) * 2.0 ** -44


def normals(seed, chain, t, D):
    """z[i], i < D: element pair p = i >> 1 takes 64 bits of the transition's stream — half (p >> 3) & 1 of block slot
    (p & 7) + 8 (p >> 4), so pairs p and p + 8 share a Philox block — and Box-Muller on a 44-bit radius uniform and a 20-bit
    angle (((wb >> 12) + 1/2) 2^-20 + 2^-53 turns: the centre of one of 2^20 cells); cosine half for even i, sine half for odd i"""
    z = np.empty(D)
    for p in range((D + 1) // 2):
        blk = stream_block(seed, chain, t, (p & 7) + 8 * (p >> 4))
        wa, wb = (blk[2], blk[3]) if (p >> 3) & 1 else (blk[0], blk[1])
        r = math.sqrt(-2.0 * math.log(u44(wa, wb)))
        a = 2.0 * math.pi * (((wb >> 12) + 0.5) * 2.0 ** -20 + 2.0 ** -53)
        z[2 * p] = r * math.cos(a)
        if 2 * p + 1 < D:
            z[2 * p + 1] = r * math.sin(a)
    return z


def accept_uniform(seed, chain, t, D):
    x, y, _, _ = stream_block(seed, chain, t, (D + 1) // 2)
    return u44(x, y)


INIT_T = (1 << 40) - 1          # transition index "-1": the initial-state stream


# ------------------------------------------------------------------ targets: (lt, grad) closures
def diag_target(w, mu, c):
    w, mu = np.asarray(w, float), np.asarray(mu, float)
    return (lambda x: c - float(np.sum(w * (x - mu) ** 2))), (lambda x: -2.0 * w * (x - mu))


def dense_target(P, mu, c):
    P, mu = np.asarray(P, float), np.asarray(mu, float)
    return (lambda x: c - 0.5 * float((x - mu) @ P @ (x - mu))), (lambda x: -(P @ (x - mu)))


def logistic_target(X, y, lam):
    X, y = np.asarray(X, float), np.asarray(y, float)
    d = X.shape[1]

    def lt(p):       # analytical.jl:11-16: ploglikelihood + plogprior
        xp = X @ p
        return float(xp @ y - np.sum(np.logaddexp(0.0, xp)) - 0.5 * (p @ p / lam + d * math.log(2.0 * math.pi * lam)))

    def grad(p):     # :17-18
        xp = X @ p
        return X.T @ (y - 1.0 / (1.0 + np.exp(-xp))) - p / lam

    return lt, grad


def hier_normal_target(Y, xc, prior_prec, ga, gb):
    """The builder-defined BUGS "Rats" target of BASELINE cfg 5 (include/klara_hip.h KLARA_TARGET_HIER_NORMAL): Y_ij ~ N(a_i + b_i xc_j, s_c^2),
    a_i ~ N(a_c, s_a^2), b_i ~ N(b_c, s_b^2), a_c, b_c ~ N(0, 1 / prior_prec), 1 / s_k^2 ~ Gamma(ga, gb), in
    theta = (a_1, b_1, ..., a_R, b_R, a_c, b_c, log s_c, log s_a, log s_b); gradient by hand from the log-density."""
    Y, xc = np.asarray(Y, float), np.asarray(xc, float)
    R, T = Y.shape

    def split(v):
        return v[0:2 * R:2], v[1:2 * R:2], v[2 * R], v[2 * R + 1], v[2 * R + 2], v[2 * R + 3], v[2 * R + 4]

    def lt(v):
        a, b, ac, bc, sc, sa, sb = split(v)
        r = Y - a[:, None] - b[:, None] * xc[None, :]
        wc, wa, wb = math.exp(-2 * sc), math.exp(-2 * sa), math.exp(-2 * sb)
        return float(-R * T * sc - 0.5 * wc * np.sum(r * r) - R * sa - 0.5 * wa * np.sum((a - ac) ** 2) - R * sb - 0.5 * wb * np.sum((b - bc) ** 2)
                     - 0.5 * prior_prec * (ac * ac + bc * bc) + sum(-2 * ga * s_ - gb * math.exp(-2 * s_) for s_ in (sc, sa, sb)))

    def grad(v):
        a, b, ac, bc, sc, sa, sb = split(v)
        r = Y - a[:, None] - b[:, None] * xc[None, :]
        wc, wa, wb = math.exp(-2 * sc), math.exp(-2 * sa), math.exp(-2 * sb)
        g = np.empty_like(v)
        g[0:2 * R:2] = wc * r.sum(axis=1) - wa * (a - ac)
        g[1:2 * R:2] = wc * (r * xc[None, :]).sum(axis=1) - wb * (b - bc)
        g[2 * R] = wa * np.sum(a - ac) - prior_prec * ac
        g[2 * R + 1] = wb * np.sum(b - bc) - prior_prec * bc
        g[2 * R + 2] = -R * T + wc * np.sum(r * r) - 2 * ga + 2 * gb * wc
        g[2 * R + 3] = -R + wa * np.sum((a - ac) ** 2) - 2 * ga + 2 * gb * wa
        g[2 * R + 4] = -R + wb * np.sum((b - bc) ** 2) - 2 * ga + 2 * gb * wb
        return g

    return lt, grad


# ------------------------------------------------------------------ one chain of one job
class Chain:
    def __init__(self, sampler, lt, grad, x0, seed, chain_id, *, sigma=None, driftstep=None, leapstep=None, nleaps=None,
                 widths=None, stepout=True, tuner="vanilla", verbose=False, targetrate=None, period=100, score_k=7.0,
                 nsteps=0, burnin=0, thinning=1, nadapt=0, eps0bar=1.0, h0bar=0.0, gamma=0.05, t0=10, kappa=0.75):
        self.sampler, self.ltf, self.gradf = sampler, lt, grad
        self.x = np.array(x0, float)
        self.D = self.x.size
        self.seed, self.cid = seed, chain_id
        self.sigma, self.widths, self.stepout = sigma, widths, stepout
        self.nleaps = nleaps
        self.tuner, self.verbose, self.targetrate, self.period, self.score_k = tuner, verbose, targetrate, period, score_k
        self.nsteps, self.burnin, self.thinning = nsteps, burnin, thinning
        # initialize!: MH.jl:72-85, MALA.jl:76-90, HMC.jl:106-120, SliceSampler.jl:40-48
        self.lt = self.ltf(self.x)
        assert math.isfinite(self.lt)
        self.g = self.gradf(self.x) if sampler in ("mala", "hmc") else None
        # tuner_state: samplers.jl:29-45 — MH's step is 1.0 and never read; totproposed starts at the period
        # (the slice sampler falls to the generic tuner_state: BasicMCTune(NaN, 0, 0, period), samplers.jl:29)
        self.step = {"mh": 1.0, "mala": driftstep, "hmc": leapstep, "slice": float("nan")}[sampler]
        self.accepted, self.proposed, self.totproposed = 0, 0, period
        # DualAveragingMCTuner (HMC only): tuner_state HMC.jl:124-133 (lambda = nleaps * leapstep, eps_bar, h_bar), mu = log(10 step)
        # HMC.jl:209.  (The reference first passes the step through initialize_step!, which cannot run for multivariate parameters —
        # samplers.jl:195 reads an undefined variable; as in the library, the sampler's leapstep is the starting step.)
        self.nadapt, self.gamma, self.t0, self.kappa = nadapt, gamma, t0, kappa
        if tuner == "da":
            self.lam, self.mu_da, self.epsbar, self.hbar = nleaps * leapstep, math.log(10.0 * leapstep), eps0bar, h0bar
        self.t = 0
        self.accepts, self.saved = [], []

    def _cnt(self):
        if self.sampler in ("mh", "slice"):
            return self.verbose                                   # iterate/MH.jl:73-75, SliceSampler.jl:61-63
        return (self.tuner in ("vanilla", "da") and self.verbose) or self.tuner == "rate"      # iterate/MALA.jl:79, HMC.jl:128-132

    def _mh(self, t):                                             # iterate/MH.jl:72-124
        xp = self.x + self.sigma * normals(self.seed, self.cid, t, self.D)          # :79
        ltp = self.ltf(xp)                                                           # :81
        ratio = ltp - self.lt                                                        # :83
        acc = ratio > 0 or ratio > math.log(accept_uniform(self.seed, self.cid, t, self.D))     # :97
        if acc:
            self.x, self.lt = xp, ltp
        return acc

    def _mala(self, t):                                           # iterate/MALA.jl:78-128
        h = self.step
        mu = self.x + 0.5 * h * self.g                                               # :83
        xp = mu + math.sqrt(h) * normals(self.seed, self.cid, t, self.D)             # :84
        ltp, gp = self.ltf(xp), self.gradf(xp)                                       # :86
        ratio = ltp - self.lt                                                        # :88
        ratio += float(np.sum(0.5 * ((mu - xp) ** 2 / h)))                           # :90
        mup = xp + 0.5 * h * gp                                                      # :91
        ratio -= float(np.sum(0.5 * ((mup - self.x) ** 2 / h)))                      # :92
        acc = ratio > 0 or ratio > math.log(accept_uniform(self.seed, self.cid, t, self.D))     # :94
        if acc:
            self.x, self.g, self.lt = xp, gp, ltp
        return acc

    def _hmc(self, t):                                            # iterate/HMC.jl:124-201
        eps = self.step
        p = normals(self.seed, self.cid, t, self.D)                                  # :135
        h0 = self.lt - 0.5 * float(p @ p)                                            # :137, samplers.jl:103
        xp, gp = self.x.copy(), self.g.copy()
        nleaps = max(1, int(round(self.lam / eps))) if self.tuner == "da" else self.nleaps     # :142-144 (round: ties to even, as Julia's)
        for _ in range(nleaps):                                                      # :146-155, samplers.jl:122-134
            p = p + 0.5 * eps * gp
            xp = xp + eps * p
            gp = self.gradf(xp)
            p = p + 0.5 * eps * gp
        ltp = self.ltf(xp)                                                           # :157
        h1 = ltp - 0.5 * float(p @ p)                                                # :159
        d = h1 - h0
        a = 1.0 if d >= 0 else math.exp(d)                                           # :163  min(1, exp(ratio))
        acc = accept_uniform(self.seed, self.cid, t, self.D) < a                     # :165  rand() always drawn
        self.a = a
        if acc:
            self.x, self.g, self.lt = xp, gp, ltp
        return acc

    def _slice(self, t):                                          # iterate/SliceSampler.jl:60-109
        for i in range(self.D):
            b0 = stream_block(self.seed, self.cid, t, i << 14)
            logu = math.log(u52(b0[0], b0[1])) + self.lt                             # :66
            r = u52(b0[2], b0[3])                                                    # :71
            w, xi = self.widths[i], self.x[i]
            lo, hi = xi - r * w, xi + (1.0 - r) * w                                  # :72-73

            def at(v):
                y = self.x.copy(); y[i] = v
                return self.ltf(y)

            if self.stepout:                                                         # :75-89
                while at(lo) > logu:
                    lo -= w
                while at(hi) > logu:
                    hi += w
            a = 1
            while True:                                                              # :91-106
                bb = stream_block(self.seed, self.cid, t, (i << 14) | ((a + 1) >> 1))       # attempts 2k - 1 and 2k share block k: words (x, y), then (z, w)
                cand = (u52(bb[0], bb[1]) if a & 1 else u52(bb[2], bb[3])) * (hi - lo) + lo      # :92-93
                lc = at(cand)                                                        # :94
                if lc > logu:
                    break
                if cand > xi:
                    hi = cand
                elif cand < xi:
                    lo = cand
                else:
                    raise RuntimeError("slice shrunk to the current point")          # :102
                a += 1
            self.x[i], self.lt = cand, lc                                            # :108
        return True

    def run(self, n):
        for _ in range(n):
            t = self.t
            cnt = self._cnt()
            if cnt:
                self.proposed += 1
            acc = {"mh": self._mh, "mala": self._mala, "hmc": self._hmc, "slice": self._slice}[self.sampler](t)
            if cnt and acc and self.sampler != "slice":
                self.accepted += 1
            if self.tuner == "da":                                                   # iterate/HMC.jl:225-249
                count = t + 1                                                        # job.sstate.count, incremented at :125-127
                if count <= self.nadapt:                                             # tune!: DualAveragingMCTuner.jl:95-101
                    hw = 1.0 / (count + self.t0)
                    self.hbar = (1.0 - hw) * self.hbar + hw * (self.targetrate - self.a)
                    self.step = math.exp(self.mu_da - math.sqrt(count) * self.hbar / self.gamma)
                    ew = count ** (-self.kappa)
                    self.epsbar = math.exp((1.0 - ew) * math.log(self.epsbar) + ew * math.log(self.step))
                    if cnt and self.proposed % self.period == 0:                     # the verbose report: rate!, reset_burnin!
                        self.totproposed += self.proposed
                        self.accepted = self.proposed = 0
                else:
                    self.step = self.epsbar                                          # :247
                self.accepts.append(bool(acc))
                i = t + 1
                if i > self.burnin and (i - self.burnin - 1) % self.thinning == 0 and i <= self.nsteps:
                    self.saved.append(self.x.copy())
                self.t += 1
                continue
            # burn-in block: iterate/MALA.jl:130-152, HMC.jl:203-224 (rate!, tune!, reset_burnin!); MH.jl:126-140 and
            # SliceSampler.jl:111-119 have the same block without tune! (their step is never adapted)
            if cnt and self.totproposed <= self.burnin and self.proposed % self.period == 0:
                if self.tuner == "rate" and self.sampler in ("mala", "hmc"):
                    rate = self.accepted / self.proposed                                          # tuners.jl:27-29
                    self.step *= 2.0 / (1.0 + math.exp(-self.score_k * (rate - self.targetrate)))  # AcceptanceRateMCTuner.jl:9,46
                self.totproposed += self.proposed
                self.accepted = self.proposed = 0
            self.accepts.append(bool(acc))
            i = t + 1                                                                # the 1-based step index of run(job)
            if i > self.burnin and (i - self.burnin - 1) % self.thinning == 0 and i <= self.nsteps:    # BasicMCRange.jl:36
                self.saved.append(self.x.copy())
            self.t += 1


def init_state_normal(seed, chain_id, D):
    return normals(seed, chain_id, INIT_T, D)
