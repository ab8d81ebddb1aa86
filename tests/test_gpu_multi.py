"""Multi-GPU test (-m gpu; skipped on a box with one GPU): min(device_count, 4) ranks, one process per GPU, the job sharded by
`chain_offset`, and the ONE exchange of the path — the all-reduce of the pooled chain summaries — through the C ABI
(klara_comm_unique_id / klara_comm_init / klara_gather_summaries: RCCL over xGMI).

Replaces `run(job::Vector) = map(run, job)` (/root/reference/src/jobs/jobs.jl:212) spread over GPUs: every chain's result must
not depend on how many GPUs ran the job, and the gathered sums must equal the single-GPU job's.
"""
import ctypes as C
import multiprocessing as mp
import os

import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("gpu_required")]

NCHAINS, NDIMS, NSTEPS, BURNIN = 4099, 100, 60, 20          # (4099: ragged shards and a ragged last wavefront group)


def _engine(K, L, nchains, offset, device):
    return K.Engine(sampler=L.SAMPLER_MALA, target=K.GaussDiagTarget.negdot(NDIMS), nchains=nchains, nsteps=NSTEPS, burnin=BURNIN,
                    driftstep=0.05, chain_offset=offset, device=device, monitor=L.MON_SUMMARIES, seed=77)


def _rank_main(rank, world, uid_q, out_q):
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    sys.path.insert(0, str(root))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import klara_jl_amd as K
    from klara_jl_amd import _lib as L
    lib = L.load()
    try:
        uid = (C.c_uint8 * 128)()
        if rank == 0:
            L.check(lib.klara_comm_unique_id(uid), "klara_comm_unique_id")
            for _ in range(world - 1):
                uid_q.put(bytes(uid))
        else:
            uid = (C.c_uint8 * 128).from_buffer_copy(uid_q.get(timeout=120))
        offset, n = K.shard_chains(NCHAINS, rank, world)
        eng = _engine(K, L, n, offset, rank)
        eng.init_state_normal()
        eng.run(NSTEPS)
        comm = C.c_void_p()
        L.check(lib.klara_comm_init(C.byref(comm), world, rank, uid, rank), "klara_comm_init")
        s = np.empty(NDIMS); q = np.empty(NDIMS)
        na, nt, ns, nc = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        L.check(lib.klara_gather_summaries(eng._h, comm, s.ctypes.data, q.ctypes.data, C.byref(na), C.byref(nt), C.byref(ns), C.byref(nc)),
                "klara_gather_summaries")
        x, lt, g = eng.state()
        out_q.put((rank, offset, n, s, q, (na.value, nt.value, ns.value, nc.value), x, lt, g, None))
        L.check(lib.klara_comm_destroy(comm), "klara_comm_destroy")
        eng.close()
    except Exception as exc:        # the parent must not wait for a rank that died
        out_q.put((rank, 0, 0, None, None, None, None, None, None, repr(exc)))


def test_sharded_job_and_rccl_summary_gather_through_the_c_abi():
    import torch
    ndev = torch.cuda.device_count()
    if ndev < 2:
        pytest.skip("needs at least 2 GPUs on the box")
    world = min(ndev, 4)
    ctx = mp.get_context("spawn")
    uid_q, out_q = ctx.Queue(), ctx.Queue()
    procs = [ctx.Process(target=_rank_main, args=(r, world, uid_q, out_q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [out_q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
    errs = [r[-1] for r in results if r[-1] is not None]
    assert not errs, errs
    results.sort(key=lambda r: r[0])

    import klara_jl_amd as K
    from klara_jl_amd import _lib as L
    ref = _engine(K, L, NCHAINS, 0, 0)
    ref.init_state_normal(); ref.run(NSTEPS)
    x, lt, g = ref.state()
    s, q, na, nt, nsaved = ref.pooled_summaries()
    assert sum(r[2] for r in results) == NCHAINS
    for rank, offset, n, gs, gq, counts, rx, rlt, rg, _ in results:
        sl = slice(offset, offset + n)
        assert np.array_equal(rx, x[sl]) and np.array_equal(rlt, lt[sl]) and np.array_equal(rg, g[sl]), rank      # bit-identical chains
        assert np.allclose(gs, s, rtol=1e-12, atol=1e-9) and np.allclose(gq, q, rtol=1e-12), rank                # gathered sums, every rank
        assert counts == (na, nt, nsaved * NCHAINS, NCHAINS), (rank, counts)
    ref.close()
