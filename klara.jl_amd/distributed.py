"""Multi-GPU: chains shard across ranks, no collective on the data path.

Chains are independent (`run(job::Vector) = map(run, job)`, src/jobs/jobs.jl:212), so rank r of R owns the
block [r*N/R, (r+1)*N/R) and seeds its Philox subsequences with the GLOBAL chain id: results do not depend
on R.  The only exchange is one end-of-run all-reduce of the pooled chain summaries
(sum x[D], sum x^2[D], n_accept, n_transitions, n_saved*chains): (2D+3) doubles ~ 1.6 kB at D = 100,
latency-bound over RCCL/xGMI.  Backend "nccl" (= RCCL) on GPUs, "gloo" in the CPU tests.
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np


def shard_chains(nchains_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Block partition -> (chain_offset, nchains_local); remainders go to the low ranks."""
    base, rem = divmod(int(nchains_total), int(world))
    count = base + (1 if rank < rem else 0)
    offset = rank * base + min(rank, rem)
    return offset, count


def allreduce_summaries(local: Dict[str, np.ndarray], group=None, device=None) -> Dict[str, np.ndarray]:
    """SUM-all-reduce of pooled summaries; `local` holds sum[D], sumsq[D], naccept, ntransitions, nsamples.

    Returns the global sums plus derived posterior moments (mean, var) and the acceptance rate.
    Works without torch.distributed initialised (single process) — then it only derives the moments.
    """
    import torch
    import torch.distributed as dist

    d = int(np.asarray(local["sum"]).size)
    buf = np.concatenate([np.asarray(local["sum"], dtype=np.float64).ravel(),
                          np.asarray(local["sumsq"], dtype=np.float64).ravel(),
                          np.array([local["naccept"], local["ntransitions"], local["nsamples"]], dtype=np.float64)])
    if dist.is_available() and dist.is_initialized():
        if device is None:
            device = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
        t = torch.from_numpy(buf).to(device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        buf = t.cpu().numpy()
    s, q = buf[:d], buf[d:2 * d]
    nacc, ntr, ns = buf[2 * d], buf[2 * d + 1], buf[2 * d + 2]
    out = {"sum": s, "sumsq": q, "naccept": nacc, "ntransitions": ntr, "nsamples": ns}
    if ns > 0:
        m = s / ns
        out["mean"] = m
        out["var"] = q / ns - m * m
    if ntr > 0:
        out["acceptance"] = nacc / ntr
    return out


def gather_engine_summaries(engine, group=None) -> Dict[str, np.ndarray]:
    """Pooled summaries of this rank's Engine, all-reduced over the job's ranks."""
    with_sums = bool(engine.monitor & 0x4)
    s, q, nacc, ntr, nsaved = engine.pooled_summaries(with_sums=with_sums)
    d = engine.ndims
    local = {"sum": s if s is not None else np.zeros(d), "sumsq": q if q is not None else np.zeros(d),
             "naccept": nacc, "ntransitions": ntr, "nsamples": nsaved * engine.nchains if with_sums else 0}
    return allreduce_summaries(local, group=group)
