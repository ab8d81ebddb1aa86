"""world_size-2 gloo test of the only collective on the path: the all-reduce of pooled chain summaries."""
import os
import socket

import numpy as np
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    sys.path.insert(0, str(root))
    import torch.distributed as dist
    import klara_jl_amd as K
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    off, cnt = K.shard_chains(10, rank, world)
    rng = np.random.default_rng(100)            # same stream on both ranks; each takes its block
    allx = rng.standard_normal((10, 50, 3))     # chains x saved steps x D
    mine = allx[off:off + cnt]
    local = {"sum": mine.sum((0, 1)), "sumsq": (mine ** 2).sum((0, 1)), "naccept": 7 * cnt,
             "ntransitions": 60 * cnt, "nsamples": 50 * cnt}
    out = K.allreduce_summaries(local)
    q.put((rank, out["mean"], out["var"], out["acceptance"], out["nsamples"]))
    dist.destroy_process_group()


def test_summary_allreduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    allx = np.random.default_rng(100).standard_normal((10, 50, 3)).reshape(-1, 3)
    for _, m, v, a, ns in res:
        assert ns == 500
        assert np.allclose(m, allx.mean(0)) and np.allclose(v, allx.var(0))
        assert a == 7 / 60


def _worker_offset(rank, world, port, q):
    """the same exchange on data with |mean| >> sd (the rats model's alpha_c: mean 242, sd 2.7): raw sums formed exactly (math.fsum)"""
    import math, sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    import torch.distributed as dist
    import klara_jl_amd as K
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x = 242.0 + 2.7 * np.random.default_rng(5).standard_normal((4000, 2)); x[:, 1] *= 1e-3
    off, cnt = K.shard_chains(4000, rank, world)
    mine = x[off:off + cnt]
    local = {"sum": np.array([math.fsum(mine[:, j]) for j in range(2)]), "sumsq": np.array([math.fsum(mine[:, j] ** 2) for j in range(2)]),
             "naccept": 1, "ntransitions": 2, "nsamples": cnt}
    out = K.allreduce_summaries(local)
    q.put((rank, out["mean"], out["var"]))
    dist.destroy_process_group()


def test_pooled_variance_of_offset_data_over_two_ranks():
    """the pooled variance of the all-reduced summaries on data whose mean is ~90 standard deviations from zero.
    Per-rank (n, mean, M2) from exact rational arithmetic, Chan's combination across the ranks: within 1e-12 (relative) of the
    variance computed in exact arithmetic from the samples — what is left is the rounding of the squares and of the raw sums."""
    from fractions import Fraction
    ctx = mp.get_context("spawn")
    q = ctx.Queue(); port = _free_port()
    procs = [ctx.Process(target=_worker_offset, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60); assert p.exitcode == 0
    x = 242.0 + 2.7 * np.random.default_rng(5).standard_normal((4000, 2)); x[:, 1] *= 1e-3
    for j in range(2):
        fx = [Fraction(float(v)) for v in x[:, j]]
        m = sum(fx) / len(fx)
        var = float(sum((v - m) ** 2 for v in fx) / len(fx))
        for _, mean, v in res:
            assert abs(mean[j] - float(m)) <= 1e-15 * abs(float(m))
            assert abs(v[j] - var) <= 1e-12 * var, (j, v[j], var)


def test_allreduce_single_process_derives_moments():
    import klara_jl_amd as K
    x = np.random.default_rng(1).standard_normal((40, 2))
    out = K.allreduce_summaries({"sum": x.sum(0), "sumsq": (x ** 2).sum(0), "naccept": 10, "ntransitions": 40,
                                 "nsamples": 40})
    assert np.allclose(out["mean"], x.mean(0)) and np.allclose(out["var"], x.var(0)) and out["acceptance"] == 0.25


class _StubCommLib:
    """Stands in for libklara_hip.so's klara_comm_* entry points (no GPU, no RCCL here): records what the bootstrap hands them."""

    def __init__(self, rank):
        self.rank, self.calls = rank, []

    def klara_comm_unique_id(self, buf):
        self.calls.append(("unique_id",))
        for i in range(128):
            buf[i] = (37 * i + 11) & 0xFF          # what "rank 0's RCCL" made up
        return 0

    def klara_comm_init(self, out, nranks, rank, uid, device):
        self.calls.append(("init", int(nranks), int(rank), bytes(uid), int(device)))
        out._obj.value = 0x1000 + rank             # (ctypes.byref(c_void_p) -> the handle the "library" returns)
        return 0

    def klara_comm_info(self, handle, n, r, d):
        _, nranks, rank, _, device = [c for c in self.calls if c[0] == "init"][-1]
        n._obj.value, r._obj.value, d._obj.value = nranks, rank, device
        return 0

    def klara_comm_destroy(self, handle):
        self.calls.append(("destroy",))
        return 0


def _worker_bootstrap(rank, world, port, q):
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    import torch.distributed as dist
    import klara_jl_amd as K
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = _StubCommLib(rank)
    comm = K.bootstrap_comm(lib, rank, world, 5 + rank, K.torch_broadcast_bytes())
    info = comm.info()
    comm.close()
    q.put((rank, lib.calls, info))
    dist.destroy_process_group()


def test_comm_bootstrap_id_broadcast_and_rank_bookkeeping_world2():
    """bench.py --gpus N makes the library's communicator through the C ABI: rank 0 alone asks for the id
    (klara_comm_unique_id), the SAME 128 bytes reach every rank over the rendezvous group, and every rank joins with its own index,
    the job's rank count and its own device — checked here with a stub in place of the library (no RCCL on the CPU box), world size 2
    over gloo; the real entry points run on the GPU box (tests/test_gpu_multi.py)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_bootstrap, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = bytes((37 * i + 11) & 0xFF for i in range(128))
    (r0, calls0, info0), (r1, calls1, info1) = res
    assert [c[0] for c in calls0] == ["unique_id", "init", "destroy"] and [c[0] for c in calls1] == ["init", "destroy"]     # only rank 0 makes the id
    assert calls0[1] == ("init", 2, 0, want, 5) and calls1[0] == ("init", 2, 1, want, 6)
    assert info0 == (2, 0, 5) and info1 == (2, 1, 6)


def test_comm_bootstrap_refuses_a_broadcast_that_lost_the_id():
    import sys
    from pathlib import Path
    import pytest
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    import klara_jl_amd as K
    lib = _StubCommLib(1)
    with pytest.raises(RuntimeError, match="did not deliver"):
        K.bootstrap_comm(lib, 1, 2, 0, lambda b: None)
    with pytest.raises(RuntimeError, match="did not deliver"):
        K.bootstrap_comm(lib, 1, 2, 0, lambda b: b"short")
    assert lib.calls == []                                   # no rank joins with a wrong id
    one = K.bootstrap_comm(_StubCommLib(0), 0, 1, 0, lambda b: b)      # one rank: its own id comes straight back
    assert one.info() == (1, 0, 0)
    one.close()
    # a rank 0 that cannot make an id still broadcasts (the other ranks are waiting for it), then every rank raises
    class NoId(_StubCommLib):
        def klara_comm_unique_id(self, buf):
            return 5                                            # KLARA_ERR_UNSUPPORTED: no RCCL on the box
    sent = []
    with pytest.raises(Exception, match="klara_comm_unique_id"):
        K.bootstrap_comm(NoId(0), 0, 2, 0, lambda b: sent.append(b) or b)
    assert sent == [b""]
    # klara_comm_init that never returns (a peer never joins): the deadline, on a helper thread
    import time
    class Hangs(_StubCommLib):
        def klara_comm_init(self, out, nranks, rank, uid, device):
            time.sleep(30)
            return 0
    t0 = time.time()
    with pytest.raises(K.CommBootstrapTimeout):
        K.bootstrap_comm(Hangs(0), 0, 2, 0, lambda b: b, timeout=0.5)
    assert time.time() - t0 < 5
    ok = K.bootstrap_comm(_StubCommLib(0), 0, 1, 3, lambda b: b, timeout=5.0)      # the same path when the call does return
    assert ok.info() == (1, 0, 3)
