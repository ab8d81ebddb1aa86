"""Multi-GPU: chains shard across ranks, no collective on the data path.

Chains are independent (`run(job::Vector) = map(run, job)`, src/jobs/jobs.jl:212), so rank r of R owns the
block [r*N/R, (r+1)*N/R) and seeds its Philox subsequences with the GLOBAL chain id: results do not depend
on R.  The only exchange is the end-of-run merge of the pooled chain summaries: every rank holds (n_r, mean_r[D], M2_r[D]) of its
chains — formed on the device without the cancellation of sumsq/n - mean^2 (klara_gather_moments, include/klara_hip.h) — and the
ranks are merged by Chan's update in three SUM all-reduces (3 counters, D weighted means, D sums of squares): ~1.6 kB at D = 100,
latency-bound over RCCL/xGMI.  Backend "nccl" (= RCCL) on GPUs, "gloo" in the CPU tests.  No rational arithmetic, no exceptions on
non-finite sums (they propagate as NaN moments on every rank alike, so no rank can leave the collectives early).
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Tuple

import numpy as np


def shard_chains(nchains_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Block partition -> (chain_offset, nchains_local); remainders go to the low ranks."""
    base, rem = divmod(int(nchains_total), int(world))
    count = base + (1 if rank < rem else 0)
    offset = rank * base + min(rank, rem)
    return offset, count


class KlaraComm:
    """The library's own communicator (klara_comm_*: RCCL loaded by the library, include/klara_hip.h) — what a Julia binding holds as
    `HIPComm`.  `Engine.pooled_moments(comm.handle)` / `klara_gather_moments(h, comm, ...)` is then the job's one exchange."""

    def __init__(self, lib, nranks: int, rank: int, uid: bytes, device: int):
        import ctypes as C
        from . import _lib as L
        if len(uid) != COMM_ID_BYTES:
            raise ValueError(f"the communicator id has {len(uid)} bytes, expected {COMM_ID_BYTES}")
        self._lib, self.handle = lib, C.c_void_p()
        buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(uid)
        L.check(lib.klara_comm_init(C.byref(self.handle), int(nranks), int(rank), buf, int(device)), "klara_comm_init")

    @staticmethod
    def unique_id(lib) -> bytes:
        """rank 0 only: RCCL's ncclUniqueId (128 bytes) that the caller ships to the other ranks"""
        import ctypes as C
        from . import _lib as L
        buf = (C.c_uint8 * COMM_ID_BYTES)()
        L.check(lib.klara_comm_unique_id(buf), "klara_comm_unique_id")
        return bytes(buf)

    def info(self) -> Tuple[int, int, int]:
        """(ranks, this rank, device) as the communicator itself reports them (ncclCommCount / ncclCommUserRank)"""
        import ctypes as C
        from . import _lib as L
        n, r, d = C.c_int32(), C.c_int32(), C.c_int32()
        L.check(self._lib.klara_comm_info(self.handle, C.byref(n), C.byref(r), C.byref(d)), "klara_comm_info")
        return n.value, r.value, d.value

    def close(self) -> None:
        if self.handle:
            self._lib.klara_comm_destroy(self.handle)
            self.handle = None


COMM_ID_BYTES = 128         # KLARA_COMM_ID_BYTES


class CommBootstrapTimeout(RuntimeError):
    """klara_comm_init did not return by the deadline; a helper thread is still inside it (the caller should leave with os._exit)"""


def bootstrap_comm(lib, rank: int, world: int, device: int, broadcast: Callable[[Optional[bytes]], bytes],
                   timeout: Optional[float] = None) -> KlaraComm:
    """One communicator over the job's ranks: rank 0 makes the id (klara_comm_unique_id), `broadcast(id on rank 0 / None elsewhere)`
    returns rank 0's bytes on every rank — the caller's transport, used for these 128 bytes and nothing else — and every rank joins
    with its own index and device (klara_comm_init is collective: it returns when all `world` ranks have called it).
    Every rank ALWAYS takes part in the broadcast, on the calling thread (a rank 0 that could not make an id broadcasts b"": all ranks
    raise together).  With `timeout`, klara_comm_init — the one call that waits for the other ranks inside RCCL — runs on a helper
    thread and CommBootstrapTimeout is raised past the deadline."""
    uid, err = None, None
    if rank == 0:
        try:
            uid = KlaraComm.unique_id(lib)
        except Exception as exc:            # the other ranks are waiting in the broadcast: tell them
            uid, err = b"", exc
    uid = broadcast(uid)
    if err is not None:
        raise err
    if uid is None or len(uid) != COMM_ID_BYTES:
        raise RuntimeError(f"rank {rank}: the broadcast did not deliver rank 0's communicator id")
    if timeout is None:
        return KlaraComm(lib, world, rank, uid, device)
    import threading
    box = {}

    def join():
        try:
            box["comm"] = KlaraComm(lib, world, rank, uid, device)
        except Exception as exc:
            box["error"] = exc
    th = threading.Thread(target=join, daemon=True)
    th.start(); th.join(timeout=timeout)
    if th.is_alive():
        raise CommBootstrapTimeout(f"rank {rank}: klara_comm_init did not return within {timeout:.0f} s (RCCL bootstrap)")
    if "error" in box:
        raise box["error"]
    return box["comm"]


def torch_broadcast_bytes(group=None) -> Callable[[Optional[bytes]], bytes]:
    """`broadcast` for bootstrap_comm over an initialised torch.distributed group (any backend; gloo in bench.py)"""
    def bc(b: Optional[bytes]) -> bytes:
        import torch.distributed as dist
        box = [b]
        dist.broadcast_object_list(box, src=0, group=group)
        return box[0]
    return bc


def gather_engine_moments_klara(engine, comm: KlaraComm) -> Dict[str, np.ndarray]:
    """The job's one exchange through the C ABI alone: klara_gather_moments(h, comm, ...) — per-chain moments pooled on the device and
    merged over the ranks by three RCCL all-reduces on the job's stream.  Same keys as gather_engine_summaries."""
    if engine.monitor & 0x4:
        mean, m2, ns, nacc, ntr, nc = engine.pooled_moments(comm.handle)
    else:                               # no running sums: the counters alone (klara_gather_summaries with NULL sums: one all-reduce)
        import ctypes as C
        from . import _lib as L
        na, nt, nsm, nch = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        L.check(engine._lib.klara_gather_summaries(engine._h, comm.handle, None, None, C.byref(na), C.byref(nt), C.byref(nsm), C.byref(nch)),
                "klara_gather_summaries")
        mean = m2 = None; ns, nacc, ntr, nc = 0, int(na.value), int(nt.value), int(nch.value)
    out = {"naccept": float(nacc), "ntransitions": float(ntr), "nsamples": float(ns), "nchains": float(nc)}
    if ns > 0:
        out.update(mean=mean, m2=m2, var=m2 / ns, sum=mean * ns, sumsq=m2 + ns * mean ** 2)
    if ntr > 0:
        out["acceptance"] = nacc / ntr
    return out


def _two_prod(a: np.ndarray, b: np.ndarray):
    """a * b = p + e exactly (Dekker / Veltkamp splitting; numpy has no fma)."""
    p = a * b
    def split(v):
        c = 134217729.0 * v                    # 2^27 + 1
        hi = c - (c - v)
        return hi, v - hi
    ah, al = split(a); bh, bl = split(b)
    e = ((ah * bh - p) + ah * bl + al * bh) + al * bl
    return p, e


def _local_moments(s: np.ndarray, q: np.ndarray, n: float):
    """(mean, M2 = sum (x - mean)^2) per dimension from raw sums with the one cancelling subtraction, q - s^2/n, carried in
    double-double: evaluated in plain doubles it loses mean^2/var digits (rats alpha_c: mean 242, sd 2.7 -> 4 digits); this way the
    only error left is the rounding already inside the raw sums themselves.  Non-finite sums give NaN / inf moments (no exception)."""
    if n <= 0:
        return np.zeros_like(s), np.zeros_like(s)
    with np.errstate(all="ignore"):
        p, pe = _two_prod(s, s)                 # s^2 = p + pe
        qh = p / n
        t, te = _two_prod(qh, np.full_like(qh, n))
        r = (p - t) - te                        # p = qh * n + r  (p - t is exact: qh * n is within an ulp of p)
        ql = (r + pe) / n
        m2 = (q - qh) - ql
        m2 = np.where(m2 < 0.0, 0.0, m2)
        return s / n, m2


def allreduce_moments(local: Dict[str, np.ndarray], group=None, device=None) -> Dict[str, np.ndarray]:
    """Chan's merge of per-rank moments: `local` holds mean[D], m2[D], nsamples, naccept, ntransitions of this rank's chains.
    Three SUM all-reduces — (nsamples, naccept, ntransitions), n_r mean_r, M2_r + n_r (mean_r - mean)^2 — every rank takes part in
    all three whatever its values are.  Works without torch.distributed initialised (single process)."""
    import torch
    import torch.distributed as dist

    mean_l = np.asarray(local["mean"], dtype=np.float64).ravel(); m2_l = np.asarray(local["m2"], dtype=np.float64).ravel()
    n_l = float(local["nsamples"])
    live = dist.is_available() and dist.is_initialized()
    if live and device is None:
        device = "cuda" if dist.get_backend(group) == "nccl" else "cpu"

    def reduce_sum(buf: np.ndarray) -> np.ndarray:
        if not live:
            return buf
        t = torch.from_numpy(np.ascontiguousarray(buf)).to(device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return t.cpu().numpy()

    cnt = reduce_sum(np.array([n_l, float(local["naccept"]), float(local["ntransitions"])], dtype=np.float64))   # (exact below 2^53)
    ns, nacc, ntr = cnt
    with np.errstate(all="ignore"):
        wsum = reduce_sum(n_l * mean_l)
        mean = wsum / ns if ns > 0 else np.zeros_like(wsum)
        m2 = reduce_sum(m2_l + n_l * (mean_l - mean) ** 2)          # Chan: within-rank + between-rank sums of squares
    out = {"naccept": nacc, "ntransitions": ntr, "nsamples": ns}
    if ns > 0:
        out["mean"] = mean
        out["m2"] = m2
        out["var"] = m2 / ns
    if ntr > 0:
        out["acceptance"] = nacc / ntr
    return out


def allreduce_summaries(local: Dict[str, np.ndarray], group=None, device=None) -> Dict[str, np.ndarray]:
    """The same exchange for a caller that holds pooled RAW sums: `local` = sum[D], sumsq[D], naccept, ntransitions, nsamples of this
    rank's chains.  Returns the global sums plus the posterior moments (mean, var) and the acceptance rate; the variance comes from
    per-rank (n, mean, M2) (_local_moments) merged by allreduce_moments, never from sumsq/n - mean^2 of the reduced sums."""
    import torch
    import torch.distributed as dist

    s_l = np.asarray(local["sum"], dtype=np.float64).ravel(); q_l = np.asarray(local["sumsq"], dtype=np.float64).ravel()
    n_l = float(local["nsamples"])
    mean_l, m2_l = _local_moments(s_l, q_l, n_l)
    out = allreduce_moments({"mean": mean_l, "m2": m2_l, "nsamples": n_l, "naccept": local["naccept"],
                             "ntransitions": local["ntransitions"]}, group=group, device=device)
    live = dist.is_available() and dist.is_initialized()
    if live:
        if device is None:
            device = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
        t = torch.from_numpy(np.concatenate([s_l, q_l])).to(device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        sq = t.cpu().numpy()
    else:
        sq = np.concatenate([s_l, q_l])
    d = s_l.size
    out["sum"], out["sumsq"] = sq[:d], sq[d:]
    return out


def gather_engine_summaries(engine, group=None) -> Dict[str, np.ndarray]:
    """Pooled summaries of this rank's Engine merged over the job's ranks: the rank's moments come from the device
    (Engine.pooled_moments -> klara_gather_moments with no communicator), the merge is allreduce_moments."""
    with_sums = bool(engine.monitor & 0x4)
    if with_sums:
        mean, m2, ns, nacc, ntr, _ = engine.pooled_moments()
    else:
        _, _, nacc, ntr, _ = engine.pooled_summaries(with_sums=False)
        mean = m2 = np.zeros(engine.ndims); ns = 0
    out = allreduce_moments({"mean": mean, "m2": m2, "nsamples": ns, "naccept": nacc, "ntransitions": ntr}, group=group)
    if "mean" in out:                       # the raw sums a caller of the previous form reads, derived (the moments are the product)
        out["sum"] = out["mean"] * out["nsamples"]
        out["sumsq"] = out["m2"] + out["nsamples"] * out["mean"] ** 2
    return out
