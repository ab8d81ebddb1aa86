"""Host-side mirror of Klara.jl's job / sampler / tuner / range API for the many-chain GPU path.

Julia is not installed in the build image, so the host side above the C ABI is Python with
the same names, argument meaning and error behaviour as the reference's constructors:

    p       = BasicContMuvParameter("p", logtarget=GaussDiagTarget.negdot(2))   # README.md:29
    model   = likelihood_model(p, False)                                        # README.md:34
    sampler = MH(np.ones(2))                                                    # README.md:38
    mcrange = BasicMCRange(nsteps=10000, burnin=1000)                           # README.md:42
    job     = BasicMCJob(model, sampler, mcrange, {"p": [5.1, -0.9]})           # README.md:50
    run(job); chain = output(job); mean(chain); acceptance(chain)               # README.md:54-66

Differences forced by the device: `logtarget=` takes a *target family object* (engine.GaussDiagTarget,
GaussDenseTarget, LogisticTarget) instead of an arbitrary closure, and `v0[key]` may be an
(nchains x D) matrix — one BasicMCJob then stands for nchains independent jobs
(`run(job::Vector) = map(run, job)`, src/jobs/jobs.jl:212, executed as one launch).
The Julia package a maintainer would ship is sketched in INTEGRATION.md.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Dict, Optional, Sequence

import numpy as np

from . import _lib as L
from .engine import CustomTarget, Engine, GaussDenseTarget, GaussDiagTarget, HierNormalTarget, LogisticTarget


# ------------------------------------------------------------------ range (src/ranges/BasicMCRange.jl)
class BasicMCRange:
    """BasicMCRange(nsteps=100, burnin=0, thinning=1) — BasicMCRange.jl:36; asserts :22-24."""

    def __init__(self, nsteps: int = 100, burnin: int = 0, thinning: int = 1):
        assert burnin >= 0, "Number of burn-in iterations should be non-negative"
        assert thinning >= 1, "Thinning should be >= 1"
        assert nsteps > burnin, "Total number of MCMC iterations should be greater than number of burn-in iterations"
        self.burnin, self.thinning = int(burnin), int(thinning)
        self.postrange = range(self.burnin + 1, int(nsteps) + 1, self.thinning)  # (burnin+1):thinning:nsteps
        # Julia's StepRange normalises `last` (BasicMCRange.jl:20: nsteps = last(postrange))
        self.nsteps = self.postrange[-1]
        self.npoststeps = len(self.postrange)

    def __repr__(self):
        return f"BasicMCRange: number of steps = {self.nsteps}, burnin = {self.burnin}, thinning = {self.thinning}"


# ------------------------------------------------------------------ samplers (src/samplers/*.jl)
class MCSampler:
    pass


class MH(MCSampler):
    """MH(sigma::Vector): symmetric, normalised MvNormal(x, sigma) random walk — MH.jl:63-66."""

    def __init__(self, sigma):
        self.sigma = np.atleast_1d(np.asarray(sigma, dtype=np.float64)).copy()
        self.symmetric, self.normalised = True, True

    kind = L.SAMPLER_MH


class MALA(MCSampler):
    """MALA(driftstep=1.) — MALA.jl:61-70."""
    kind = L.SAMPLER_MALA

    def __init__(self, driftstep: float = 1.0):
        assert driftstep > 0, "Drift step is not positive"
        self.driftstep = float(driftstep)


class HMC(MCSampler):
    """HMC(leapstep=0.1, nleaps=10) — HMC.jl:89-100."""
    kind = L.SAMPLER_HMC

    def __init__(self, leapstep: float = 0.1, nleaps: int = 10):
        assert leapstep > 0, "Leapfrog step is not positive"
        assert nleaps > 0, "Number of leapfrog steps is not positive"
        self.leapstep, self.nleaps = float(leapstep), int(nleaps)


class SliceSampler(MCSampler):
    """SliceSampler(widths, stepout=true) / SliceSampler(width=1., n=1, stepout=true) — SliceSampler.jl:22-34."""
    kind = L.SAMPLER_SLICE

    def __init__(self, widths=1.0, n: Optional[int] = None, stepout: bool = True):
        if np.ndim(widths) == 0:
            widths = np.full(int(n) if n is not None else 1, float(widths))
        self.widths = np.asarray(widths, dtype=np.float64).copy()
        assert np.all(self.widths > 0), "All widths must be positive"
        self.stepout = bool(stepout)


# ------------------------------------------------------------------ tuners (src/tuners/*.jl)
class MCTuner:
    pass


class VanillaMCTuner(MCTuner):
    """VanillaMCTuner(period=100, verbose=false) — VanillaMCTuner.jl; defaults test/VanillaMCTuner.jl:6-9."""
    kind = L.TUNER_VANILLA

    def __init__(self, period: int = 100, verbose: bool = False):
        assert period > 0, "Tuning period should be positive"
        self.period, self.verbose = int(period), bool(verbose)


def logistic(x, l=1.0, k=1.0, x0=0.0, y0=0.0):
    """stats/logistic.jl:11 (host convenience; the device uses its own deterministic exp)."""
    return l / (1.0 + np.exp(-k * (x - x0))) + y0


def logistic_rate_score(x, k=7.0):
    """AcceptanceRateMCTuner.jl:9."""
    return logistic(x, 2.0, k, 0.0, 0.0)


def erf_rate_score(x, k=3.0):
    """AcceptanceRateMCTuner.jl:17."""
    from math import erf
    return erf(k * x) + 1.0


class AcceptanceRateMCTuner(MCTuner):
    """AcceptanceRateMCTuner(targetrate; score=logistic_rate_score, period=100, verbose=false) — :38-44.

    `mode="per_chain"` keeps the reference's one-tune-per-job semantics; `mode="pooled"` shares one step
    per GPU with the acceptance rate pooled over that GPU's chains (BASELINE cfg 5).
    `score` is logistic_rate_score or erf_rate_score; `score_k` is its steepness (defaults 7 and 3).
    """
    kind = L.TUNER_ACCEPT_RATE

    def __init__(self, targetrate: float, score=logistic_rate_score, period: int = 100, verbose: bool = False,
                 score_k: float = 7.0, mode: str = "per_chain"):
        assert 0 < targetrate < 1, "Target acceptance rate should be between 0 and 1"
        assert period > 0, "Tuning period should be positive"
        if score not in (logistic_rate_score, erf_rate_score):
            raise NotImplementedError("device scores: logistic_rate_score, erf_rate_score")
        if score is erf_rate_score and score_k == 7.0:
            score_k = 3.0                       # erf_rate_score's own default steepness (AcceptanceRateMCTuner.jl:17)
        self.score_kind = 1 if score is erf_rate_score else 0
        assert mode in ("per_chain", "pooled")
        self.targetrate, self.period, self.verbose = float(targetrate), int(period), bool(verbose)
        self.score_k, self.mode = float(score_k), mode


class DualAveragingMCTuner(MCTuner):
    """DualAveragingMCTuner(targetrate, nadapt; ε0bar=1., h0bar=0., γ=0.05, t0=10, κ=0.75, period=100, verbose=false)
    — DualAveragingMCTuner.jl:54-93.  HMC only (as in the reference: HMC.jl:124-133)."""
    kind = L.TUNER_DUAL_AVERAGING

    def __init__(self, targetrate: float, nadapt: int, eps0bar: float = 1.0, h0bar: float = 0.0, gamma: float = 0.05,
                 t0: int = 10, kappa: float = 0.75, period: int = 100, verbose: bool = False):
        assert 0 < targetrate < 1, "Target acceptance rate should be between 0 and 1"
        assert nadapt > 0, "Number of adaptation steps should be positive"
        assert eps0bar > 0, "ε0bar should be positive"
        assert period > 0, "Period over which acceptance rate is reported in verbose mode should be positive"
        assert t0 > 0, "t0 should be positive"
        self.targetrate, self.nadapt, self.eps0bar, self.h0bar = float(targetrate), int(nadapt), float(eps0bar), float(h0bar)
        self.gamma, self.t0, self.kappa, self.period, self.verbose = float(gamma), int(t0), float(kappa), int(period), bool(verbose)


# ------------------------------------------------------------------ parameter / model
class BasicContMuvParameter:
    """BasicContMuvParameter(key; logtarget=...) — BasicContMuvParameter.jl:383-411.

    `logtarget` is a device target family; for LogisticTarget it plays the role of
    loglikelihood+logprior+gradlogtarget of doc/examples/swiss/MALA/analytical.jl:20-26.
    """

    def __init__(self, key: str, logtarget=None, loglikelihood: Optional[str] = None, logprior: Optional[str] = None,
                 gradloglikelihood: Optional[str] = None, gradlogprior: Optional[str] = None, ndims: Optional[int] = None, data=None,
                 **unsupported):
        if unsupported:
            raise NotImplementedError(f"closure fields not available on device: {sorted(unsupported)}")
        if logtarget is None:
            # likelihood + prior closures (BasicContMuvParameter.jl:174-201): C text for each, composed on device as
            # logtarget = loglikelihood + logprior, gradlogtarget = gradloglikelihood + gradlogprior
            if loglikelihood is None or logprior is None:
                raise ValueError("give logtarget (a target family object) or both loglikelihood and logprior (C source text)")
            if ndims is None:
                raise ValueError("ndims is required with loglikelihood / logprior closures")
            if (gradloglikelihood is None) != (gradlogprior is None):
                raise ValueError("gradloglikelihood and gradlogprior go together")
            logtarget = CustomTarget.likelihood_prior(ndims, loglikelihood, logprior, gradloglikelihood or "", gradlogprior or "", data)
        elif loglikelihood is not None or logprior is not None:
            raise ValueError("logtarget and loglikelihood / logprior are alternatives")
        if not isinstance(logtarget, (GaussDiagTarget, GaussDenseTarget, LogisticTarget, HierNormalTarget, CustomTarget)):
            raise TypeError("logtarget must be GaussDiagTarget, GaussDenseTarget, LogisticTarget, HierNormalTarget or CustomTarget")
        self.key = str(key).lstrip(":")
        self.target = logtarget


@dataclass
class GenericModel:
    """Just enough of src/models/GenericModel.jl to route v0[key] to the parameter (BasicMCJob.jl:156-185)."""
    vertices: list
    ofkey: Dict[str, int] = field(default_factory=dict)


def likelihood_model(p, isindexed: bool = True) -> GenericModel:
    """likelihood_model(vs, isindexed) — src/models/generators.jl:5-18 (single-parameter form)."""
    vs = list(p) if isinstance(p, (list, tuple)) else [p]
    return GenericModel(vs, {v.key: i for i, v in enumerate(vs) if hasattr(v, "key")})


# ------------------------------------------------------------------ chain container
class MuvChains:
    """Output of a job: the BasicContMuvParameterNState of every chain
    (src/nstates/ParameterNStates/BasicContMuvParameterNState.jl:1-21).

    value(c)            (ndims x n) column-major matrix of chain c (requires :value monitored)
    diagnosticvalues    (nsaved_steps x nchains) accept flags over the postrange when :accept requested
    """

    def __init__(self, job: "BasicMCJob"):
        self._job = job
        eng = job.engine
        self.size, self.nchains = eng.ndims, eng.nchains
        self.n = job.range.npoststeps
        self.diagnostickeys = list(job.outopts.get("diagnostics", []))
        self._sums = eng.chain_sums() if eng.monitor & L.MON_SUMMARIES else None
        self._acc = None
        if eng.monitor & L.MON_ACCEPT:
            m = eng.accept_mask()
            post = np.asarray(job.range.postrange) - 1
            self._acc = m[post[post < m.shape[0]]]

    def value(self, chain: int = 0) -> np.ndarray:
        return self._job.engine.chain(chain)

    def logtarget(self, chain: int = 0) -> np.ndarray:
        return self._job.engine.chain_fields(chain, True, False)[0]

    def gradlogtarget(self, chain: int = 0) -> np.ndarray:
        return self._job.engine.chain_fields(chain, False, True)[1]

    def loglikelihood(self, chain: int = 0) -> np.ndarray:
        return self._job.engine.chain_likelihood_prior(chain)[0]

    def logprior(self, chain: int = 0) -> np.ndarray:
        return self._job.engine.chain_likelihood_prior(chain)[1]

    @property
    def diagnosticvalues(self):
        return self._acc


def mean(chains: MuvChains, chain: Optional[int] = None) -> np.ndarray:
    """mean(s::VariableNState{Multivariate}) — stats/mean.jl:7-11: per-dimension mean over saved steps.
    chain=None returns (nchains x D) from the on-device running sums; chain=c reads that chain's history."""
    if chains._sums is None:
        if not (chains._job.engine.monitor & L.MON_HISTORY):
            raise ValueError("mean needs the running sums (summaries=True) or the stored values (monitor value)")
        if chain is not None:
            return chains.value(chain).mean(axis=1)
        return np.stack([chains.value(c).mean(axis=1) for c in range(chains.nchains)])
    s, _, n = chains._sums
    m = s / n
    return m if chain is None else m[chain]


def mcvar_iid(chains: MuvChains) -> np.ndarray:
    """mcvar(s, Val{:iid}) = var(v)/length(v) — stats/variance/mcvar.jl:5 (per chain, per dimension)."""
    if chains._sums is None:
        if not (chains._job.engine.monitor & L.MON_HISTORY):
            raise ValueError("mcvar_iid needs the running sums (summaries=True) or the stored values (monitor value)")
        return chain_mcvar(chains, "iid")
    s, q, n = chains._sums
    var = (q - s * s / n) / (n - 1)
    return var / n


def chain_mcvar(chains: MuvChains, vtype: str = "imse", batchlen: int = 100, maxlag: Optional[int] = None) -> np.ndarray:
    """mcvar(s, Val{vtype}) for EVERY chain and dimension at once, computed on device over the stored history
    (stats/variance/mcvar.jl:5,35-41,75-105,137-158) — or, for "bm" / "imse" / "ipse", from what a job with bm_batchlen / acov_maxlag
    accumulated while sampling.  Returns (nchains x D); vtype in {"iid", "bm", "imse", "ipse"}.
    `maxlag` left at its default means the reference's default (0 = n - 1 lags, mcvar.jl:75) when the values are stored, and the job's own
    window `acov_maxlag` when the job streams its autocovariances without a value history — the estimate is then TRUNCATED at that lag
    (<= 31); an explicit maxlag other than the window is refused for such a job."""
    job = chains._job
    if vtype == "bm" and job.bm_batchlen == batchlen and not (job.engine.monitor & L.MON_HISTORY):
        return job.engine.chain_bm()[0]            # streaming batch means: no history was stored
    streamed = job.acov_maxlag > 0 and not (job.engine.monitor & L.MON_HISTORY)
    if vtype in ("imse", "ipse") and streamed:
        # The autocovariances kept while sampling stop at lag acov_maxlag (<= 127); the reference's maxlag = 0 means n - 1
        # (mcvar.jl:75).  The streamed estimator is therefore only returned for the lag window it was built for — asked for
        # explicitly, maxlag == acov_maxlag — and anything else needs the stored values (a slowly mixing chain whose
        # Geyer sequence has not turned non-positive by lag 31 would be silently underestimated).
        if maxlag is not None and maxlag != job.acov_maxlag:
            raise ValueError(f"this job keeps streaming autocovariances up to lag {job.acov_maxlag} and no value history: "
                             f"mcvar(:{vtype}) is available for maxlag={job.acov_maxlag} only (asked for maxlag={maxlag}, where 0 means n - 1)")
        imse, ipse, _ = job.engine.chain_acov_mcvar(want=(vtype,))
        return imse if vtype == "imse" else ipse
    maxlag = 0 if maxlag is None else maxlag
    if vtype == "ipse":
        return job.engine.chain_mcvar_ipse(maxlag)
    return job.engine.chain_mcvar(batchlen, maxlag, want=(vtype,))[{"iid": 0, "bm": 1, "imse": 2}[vtype]]


def _mcvar_pair(chains: MuvChains, vtype: str, batchlen: int, maxlag: Optional[int]):
    """(mcvar_iid, mcvar_vtype) of every chain and dimension; vtype in {"bm", "imse", "ipse"}."""
    if vtype not in ("bm", "imse", "ipse"):
        raise ValueError(f"vtype must be 'bm', 'imse' or 'ipse', not {vtype!r}")
    has_history = bool(chains._job.engine.monitor & L.MON_HISTORY)      # (two-pass variance over the stored values when they exist)
    return (chain_mcvar(chains, "iid") if has_history else mcvar_iid(chains)), chain_mcvar(chains, vtype, batchlen, maxlag)


def chain_ess(chains: MuvChains, vtype: str = "imse", batchlen: int = 100, maxlag: Optional[int] = None) -> np.ndarray:
    """ess(s, vtype) = n * mcvar_iid / mcvar_vtype (stats/convergence/ess.jl:3) for every chain and dimension."""
    iid, v = _mcvar_pair(chains, vtype, batchlen, maxlag)
    return chains.n * iid / v


def chain_iact(chains: MuvChains, vtype: str = "imse", batchlen: int = 100, maxlag: Optional[int] = None) -> np.ndarray:
    """iact(s, vtype) = mcvar_vtype / mcvar_iid (stats/convergence/iact.jl:3) for every chain and dimension."""
    iid, v = _mcvar_pair(chains, vtype, batchlen, maxlag)
    return v / iid


def acceptance(chains: MuvChains, diagnostics: bool = True) -> np.ndarray:
    """acceptance(s::MultivariateParameterNState; key=:accept) — stats/acceptance.jl:28-34:
    mean of the accept diagnostics over the saved steps (per chain)."""
    if chains._acc is None:
        raise ValueError("request outopts diagnostics=['accept'] to record the accept diagnostics")
    return chains._acc.mean(axis=0)


# ------------------------------------------------------------------ random streams of jobs
# Determinism contract.  A job's noise is a pure function of (seed, global chain id, transition index) — Philox4x32-10, key =
# seed (csrc/detmath.h) — so an explicit `seed=` reproduces a job bit for bit on any number of GPUs.  Klara's own jobs draw from
# Julia's global, unseeded generator: two jobs built the same way are independent (`run([job1, job2])`, jobs.jl:212).  To keep
# that, a job built WITHOUT `seed=` takes a fresh key: splitmix64 of (a per-process random base from os.urandom + a counter), so
# chain c of one job never shares its stream with chain c of another; `job.seed` reports the key that was used.  The keys are
# HASHED, not an arithmetic progression: `reset(job)` moves a job to seed + k * KLARA_EPOCH_KEY_STRIDE (include/klara_hip.h), and a
# progression of default keys with that same stride would make job i after k resets replay job i + k.
_SEED_BASE = int.from_bytes(os.urandom(8), "little")
_seed_counter = 0
_M64 = 0xFFFFFFFFFFFFFFFF


def _splitmix64(z: int) -> int:
    z = (z + 0x9E3779B97F4A7C15) & _M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def _next_job_seed() -> int:
    global _seed_counter
    _seed_counter += 1
    return _splitmix64((_SEED_BASE + _seed_counter) & _M64)


# ------------------------------------------------------------------ job (src/jobs/BasicMCJob.jl)
class BasicMCJob:
    """BasicMCJob(model, sampler, range, v0; tuner=VanillaMCTuner(), outopts=...) — BasicMCJob.jl:107-185.

    outopts keys follow jobs.jl:9-43: destination in {"nstate", "none"}, monitor (["value"]),
    diagnostics ([] or ["accept"]).  Extra keyword arguments pick the shard: `chain_offset` (global id of
    the first chain), `device`, `seed` (default: a fresh key per job, see "random streams of jobs" above), `steps_per_launch`; `bm_batchlen` > 0 keeps streaming batch means so that
    `chain_mcvar(chain, "bm", bm_batchlen)` needs no stored history (destination "none"); `acov_maxlag` > 0 does the same for
    `chain_mcvar(chain, "imse" | "ipse", maxlag=acov_maxlag)` (autocovariances accumulated while sampling).
    """

    def __init__(self, model: GenericModel, sampler: MCSampler, mcrange: BasicMCRange, v0: Dict[str, Sequence],
                 tuner: Optional[MCTuner] = None, outopts: Optional[dict] = None, *, seed: Optional[int] = None,
                 chain_offset: int = 0, device: int = 0, steps_per_launch: int = 0, summaries: bool = True,
                 bm_batchlen: int = 0, acov_maxlag: int = 0, sparse_moves: int = 0):
        self.model, self.sampler, self.range = model, sampler, mcrange
        self.seed = _next_job_seed() if seed is None else int(seed)
        seed = self.seed
        self.tuner = tuner if tuner is not None else VanillaMCTuner()
        self.outopts = dict(outopts) if outopts is not None else {}
        self.outopts.setdefault("destination", "nstate")              # jobs.jl:10
        if self.outopts["destination"] != "none":
            self.outopts.setdefault("monitor", ["value"])             # jobs.jl:13-15
            self.outopts.setdefault("diagnostics", [])                # jobs.jl:37-39
        if self.outopts["destination"] not in ("nstate", "iostream", "none"):
            raise ValueError("outopts destination must be nstate, iostream or none")
        if self.outopts["destination"] == "iostream":                  # jobs.jl:17-29
            self.outopts.setdefault("filepath", "")
            self.outopts.setdefault("filesuffix", "csv")
            self.outopts.setdefault("flush", False)
            # saved steps held on the device between two writes (a ring of that many columns): the sink streams, as the
            # reference's does (one write per saved step), instead of keeping the whole run in memory
            self.outopts.setdefault("chunk", 256)
        params = [v for v in model.vertices if isinstance(v, BasicContMuvParameter)]
        if len(params) != 1:
            raise ValueError("model must hold exactly one BasicContMuvParameter")
        self.parameter = params[0]
        x0 = np.asarray(v0[self.parameter.key] if self.parameter.key in v0 else v0[":" + self.parameter.key],
                        dtype=np.float64)
        x0 = np.atleast_2d(x0)
        nchains, ndims = x0.shape
        if ndims != self.parameter.target.ndims:
            raise ValueError("v0 has the wrong number of dimensions for the target")
        monitor = 0
        if "accept" in self.outopts.get("diagnostics", []):
            monitor |= L.MON_ACCEPT
        if self.outopts["destination"] in ("nstate", "iostream"):
            mon = [str(m).lstrip(":") for m in self.outopts.get("monitor", [])]
            known = {"value", "logtarget", "gradlogtarget", "loglikelihood", "logprior"}
            if set(mon) - known:
                raise NotImplementedError(f"monitor fields not kept on device: {sorted(set(mon) - known)}")
            if "value" in mon:
                monitor |= L.MON_HISTORY
            if "logtarget" in mon:
                monitor |= L.MON_HIST_LT
            if "gradlogtarget" in mon:
                monitor |= L.MON_HIST_GRAD
            if "loglikelihood" in mon or "logprior" in mon:          # iterate/MALA.jl:104-109: kept when the parameter has the closures
                if not getattr(self.parameter.target, "has_parts", False):
                    raise ValueError("loglikelihood / logprior can only be monitored for a parameter built from those closures")
                monitor |= L.MON_HIST_LLLP
        if summaries:
            monitor |= L.MON_SUMMARIES
        kw = dict(sampler=sampler.kind, target=self.parameter.target, nchains=nchains, nsteps=mcrange.nsteps,
                  burnin=mcrange.burnin, thinning=mcrange.thinning, tuner=self.tuner.kind,
                  period=self.tuner.period, verbose=self.tuner.verbose, seed=seed, chain_offset=chain_offset,
                  device=device, monitor=monitor, steps_per_launch=steps_per_launch, bm_batchlen=int(bm_batchlen),
                  acov_maxlag=int(acov_maxlag), sparse_moves=int(sparse_moves),
                  hist_ring_cols=int(self.outopts["chunk"]) if self.outopts["destination"] == "iostream" else 0)
        self.bm_batchlen, self.acov_maxlag = int(bm_batchlen), int(acov_maxlag)
        if isinstance(sampler, MH):
            kw["mh_sigma"] = sampler.sigma
        elif isinstance(sampler, MALA):
            kw["driftstep"] = sampler.driftstep
        elif isinstance(sampler, HMC):
            kw["leapstep"], kw["nleaps"] = sampler.leapstep, sampler.nleaps
        elif isinstance(sampler, SliceSampler):
            kw["slice_widths"], kw["slice_stepout"] = sampler.widths, sampler.stepout
        if isinstance(self.tuner, DualAveragingMCTuner):
            if not isinstance(sampler, HMC):
                raise NotImplementedError("DualAveragingMCTuner is wired into HMC only (HMC.jl:124-133)")
            kw.update(targetrate=self.tuner.targetrate, da_nadapt=self.tuner.nadapt, da_eps0bar=self.tuner.eps0bar,
                      da_h0bar=self.tuner.h0bar, da_gamma=self.tuner.gamma, da_t0=self.tuner.t0, da_kappa=self.tuner.kappa)
        if isinstance(self.tuner, AcceptanceRateMCTuner):
            kw["targetrate"], kw["score_k"] = self.tuner.targetrate, self.tuner.score_k
            kw["tuner_score"] = self.tuner.score_kind
            kw["tuner_mode"] = L.TUNE_POOLED if self.tuner.mode == "pooled" else L.TUNE_PER_CHAIN
        self.engine = Engine(**kw)
        # initialize!(pstate, parameter, sampler, outopts): BasicMCJob.jl:73 (finite asserts on device)
        try:
            self.engine.set_state(x0)
        except L.KlaraError as e:
            if e.status == L.ERR_NONFINITE_INIT:
                raise AssertionError("Log-target (or its gradient) not finite: initial values out of support") from e
            raise
        self._ran = False

    def close(self):
        self.engine.close()


def run(job):
    """run(job::BasicMCJob) — BasicMCJob.jl:212-244; run(jobs::Vector) = map(run, jobs) — jobs.jl:212."""
    if isinstance(job, (list, tuple)):
        return [run(j) for j in job]
    if job.outopts["destination"] == "iostream":
        _run_to_iostream(job)
    else:
        job.engine.run(job.range.nsteps)
    job._ran = True
    return job


def _run_to_iostream(job: "BasicMCJob") -> None:
    """:destination => :iostream — CSV files per monitored field (jobs.jl:193-202; BasicContParamIOStream.jl:152-159), written
    WHILE the job runs: the device keeps a ring of outopts["chunk"] saved steps, the loop below runs until the ring is full (or
    the job is done), appends those steps to every chain's files and, with :flush, flushes them (jobs.jl:17-29) — the host
    memory a long job needs does not grow with its length (only the rows of the newly saved steps of the accept diagnostics are read
    back, klara_get_accept_rows; the device keeps one byte per transition and chain of them)."""
    from .iostream import ChainWriter
    eng = job.engine
    base = job.outopts.get("filepath", "") or "."
    suffix = job.outopts.get("filesuffix", "csv")
    chunk, thin = int(job.outopts["chunk"]), job.range.thinning
    has_v, has_lt, has_g = bool(eng.monitor & L.MON_HISTORY), bool(eng.monitor & L.MON_HIST_LT), bool(eng.monitor & L.MON_HIST_GRAD)
    has_acc, has_lllp = bool(eng.monitor & L.MON_ACCEPT), bool(eng.monitor & L.MON_HIST_LLLP)
    width = len(str(eng.nchains))
    writers = [ChainWriter(base if eng.nchains == 1 else os.path.join(base, f"chain_{c + 1:0{width}d}"), suffix, has_v, has_lt, has_g, has_acc, has_lllp)
               for c in range(eng.nchains)]
    post = np.asarray(job.range.postrange) - 1                    # 0-based transition index of every saved step
    written, done = 0, 0
    try:
        while done < job.range.nsteps:
            # transitions until `chunk` more steps have been saved (none are saved during burn-in)
            k = min(job.range.nsteps - done, max(1, (job.range.burnin - done) if done < job.range.burnin else chunk * thin))
            eng.run(k); done += k
            new = eng.saved_steps() - written
            if new <= 0:
                continue
            assert new <= chunk
            acc = None
            if has_acc:      # only the rows of the newly saved steps come back (not every transition since the start)
                first, last = int(post[written]), int(post[written + new - 1])
                acc = eng.accept_rows(first, last - first + 1)[::thin]
            for c, w in enumerate(writers):
                value = eng.chain(c)[:, -new:] if has_v else None
                lt, g = eng.chain_fields(c, has_lt, has_g) if (has_lt or has_g) else (None, None)
                ll, lp = eng.chain_likelihood_prior(c) if has_lllp else (None, None)
                w.append(value, None if lt is None else lt[-new:], None if g is None else g[:, -new:], None if acc is None else acc[:, c],
                         None if ll is None else ll[-new:], None if lp is None else lp[-new:])
                if job.outopts.get("flush", False):
                    w.flush()
            written += new
    finally:
        for w in writers:
            w.close()


def output(job: BasicMCJob) -> MuvChains:
    """output(job) — BasicMCJob.jl:279."""
    return MuvChains(job)


def reset(job: BasicMCJob, x=None):
    """reset(job[, x]) — BasicMCJob.jl:187-201."""
    if x is not None:
        x = np.atleast_2d(np.asarray(x, dtype=np.float64))
    job.engine.reset(x)
    job._ran = False
    return job
