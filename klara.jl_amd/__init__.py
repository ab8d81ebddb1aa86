"""klara_jl_amd — MI355X-native many-chain MCMC transition path behind Klara.jl's job/sampler API.

The package directory is `klara.jl_amd/` (not importable by that name); import it as
`klara_jl_amd` through the loader module at the repository root.
"""
from . import _lib  # noqa: F401
from ._lib import KlaraError  # noqa: F401
from .engine import CustomTarget, Engine, GaussDenseTarget, GaussDiagTarget, HierNormalTarget, LogisticTarget  # noqa: F401
from .api import (  # noqa: F401
    HMC, MALA, MH, AcceptanceRateMCTuner, DualAveragingMCTuner, BasicContMuvParameter, BasicMCJob, BasicMCRange, GenericModel,
    MuvChains, SliceSampler, VanillaMCTuner, acceptance, chain_ess, chain_iact, chain_mcvar, erf_rate_score, likelihood_model, logistic, logistic_rate_score,
    mcvar_iid, mean, output, reset, run,
)
from .distributed import (CommBootstrapTimeout, KlaraComm, allreduce_moments, allreduce_summaries, bootstrap_comm, gather_engine_moments_klara,  # noqa: F401
                          gather_engine_summaries, shard_chains, torch_broadcast_bytes)  # noqa: F401
from .build import build_library, build_oracle  # noqa: F401
from . import stats  # noqa: F401
from .stats import ess, iact, mcse, mcvar  # noqa: F401
from .iostream import ContMuvMarkovChain, read_chain  # noqa: F401

__all__ = [n for n in dir() if not n.startswith("_")]
