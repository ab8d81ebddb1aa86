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


def _gloo_rank_main(rank, world, port, out_q):
    """one rank of a 2-rank job: a real Engine shard on cuda:0, the summaries all-reduced by torch.distributed (gloo)"""
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    try:
        import torch.distributed as dist
        import klara_jl_amd as K
        from klara_jl_amd import _lib as L
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        offset, n = K.shard_chains(NCHAINS, rank, world)
        eng = _engine(K, L, n, offset, 0)
        eng.init_state_normal(); eng.run(NSTEPS)
        out = K.gather_engine_summaries(eng)                # the path's one exchange (distributed.py), here over gloo
        x, lt, g = eng.state()
        out_q.put((rank, offset, n, out, x, lt, g, None))
        eng.close(); dist.destroy_process_group()
    except Exception as exc:
        out_q.put((rank, 0, 0, None, None, None, None, repr(exc)))


def test_two_engine_shards_on_one_device_over_gloo():
    """the N > 1 path with REAL engines on a one-GPU box — two processes, each a shard of the job (global chain ids
    through chain_offset) on cuda:0, summaries all-reduced over gloo: per-chain states bit-identical to the unsharded job, all-reduced
    sums and moments equal to the unsharded job's to 1e-12."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    out_q = ctx.Queue()
    procs = [ctx.Process(target=_gloo_rank_main, args=(r, 2, port, out_q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [out_q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
    errs = [r[-1] for r in results if r[-1] is not None]
    assert not errs, errs
    import klara_jl_amd as K
    from klara_jl_amd import _lib as L
    ref = _engine(K, L, NCHAINS, 0, 0)
    ref.init_state_normal(); ref.run(NSTEPS)
    x, lt, g = ref.state()
    whole = K.gather_engine_summaries(ref)                  # (no process group in this process: the unsharded job's own moments)
    assert sum(r[2] for r in results) == NCHAINS
    for rank, offset, n, out, rx, rlt, rg, _ in results:
        sl = slice(offset, offset + n)
        assert np.array_equal(rx, x[sl]) and np.array_equal(rlt, lt[sl]) and np.array_equal(rg, g[sl]), rank
        assert out["nsamples"] == whole["nsamples"] == (NSTEPS - BURNIN) * NCHAINS and out["naccept"] == whole["naccept"]
        assert np.allclose(out["sum"], whole["sum"], rtol=1e-12, atol=1e-9) and np.allclose(out["sumsq"], whole["sumsq"], rtol=1e-12)
        assert np.allclose(out["mean"], whole["mean"], rtol=1e-12, atol=1e-13) and np.allclose(out["var"], whole["var"], rtol=1e-12)
    ref.close()


RATS_CHAINS, RATS_STEPS, RATS_BURNIN = 2051, 300, 100


def _rats_engine(K, L, nchains, offset):
    """HMC on the hierarchical rats model (cfg 5's shape): alpha_c has mean ~242 and sd ~2.7 — the dimension on which q/n - mean^2 loses 4 digits"""
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent))
    import cases
    t = cases.rats_target()
    x0 = t.least_squares_start()[None, :] + 0.05 * np.random.default_rng(12).standard_normal((RATS_CHAINS, t.ndims))
    eng = K.Engine(sampler=L.SAMPLER_HMC, target=t, nchains=nchains, nsteps=RATS_STEPS, burnin=RATS_BURNIN, leapstep=0.02, nleaps=8,
                   chain_offset=offset, monitor=L.MON_SUMMARIES, seed=5)
    eng.set_state(x0[offset:offset + nchains])
    return eng


def _exact_pooled_moments(s, q, nsaved):
    """pooled mean and variance from per-chain raw sums in exact rational arithmetic (the yardstick: what the sums hold)"""
    from fractions import Fraction
    n = Fraction(int(nsaved) * s.shape[0])
    mean, var = [], []
    for j in range(s.shape[1]):
        S = sum(Fraction(float(v)) for v in s[:, j]); Q = sum(Fraction(float(v)) for v in q[:, j])
        m = S / n
        mean.append(float(m)); var.append(float(Q / n - m * m))
    return np.array(mean), np.array(var)


def _gloo_rats_rank_main(rank, world, port, out_q):
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    try:
        import torch.distributed as dist
        import klara_jl_amd as K
        from klara_jl_amd import _lib as L
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        offset, n = K.shard_chains(RATS_CHAINS, rank, world)
        eng = _rats_engine(K, L, n, offset)
        eng.run(RATS_STEPS)
        out = K.gather_engine_summaries(eng)                # device-formed (n, mean, M2) of the shard, Chan's merge over gloo
        s, q, nsaved = eng.chain_sums()
        out_q.put((rank, offset, n, out, s, q, nsaved, None))
        eng.close(); dist.destroy_process_group()
    except Exception as exc:
        out_q.put((rank, 0, 0, None, None, None, 0, repr(exc)))


def test_pooled_moments_without_cancellation_one_handle_rccl_single_rank_and_two_shards(klib):
    """the pooled variance of the rats model's alpha_c (mean 242, sd 2.7) from the C path — klara_gather_moments:
    per-chain (n, mean, M2) on the device with q - s^2/n in double-double, Chan's merge over chains and blocks, and over ranks either by three
    RCCL all-reduces (comm) or by the caller's transport (comm = NULL + distributed.allreduce_moments) — within 1e-12 of the variance computed
    in exact rational arithmetic from the per-chain sums, for one handle, through a one-rank RCCL communicator, and for two real engine shards
    on one device merged over gloo."""
    import socket
    import klara_jl_amd as K
    from klara_jl_amd import _lib as L
    ref = _rats_engine(K, L, RATS_CHAINS, 0)
    ref.run(RATS_STEPS)
    s, q, nsaved = ref.chain_sums()
    assert nsaved == RATS_STEPS - RATS_BURNIN
    emean, evar = _exact_pooled_moments(s, q, nsaved)
    D = ref.ndims
    ac = D - 5                                              # alpha_c (theta = alpha_1, beta_1, ..., alpha_c, beta_c, log sigma_c, ...)
    assert 230 < emean[ac] < 255 and evar[ac] < 100
    mean, m2, ns, na, nt, nc = ref.pooled_moments()
    assert (ns, nc, nt) == (nsaved * RATS_CHAINS, RATS_CHAINS, RATS_STEPS * RATS_CHAINS)
    assert np.allclose(mean, emean, rtol=1e-14, atol=1e-16)
    assert np.all(np.abs(m2 / ns - evar) <= 1e-12 * evar), np.max(np.abs(m2 / ns - evar) / evar)
    ps, pq, pna, _, _ = ref.pooled_summaries()
    assert na == pna
    # through RCCL (one rank: the call sequence, the packing and the between-rank kernel), twice: a repeatable collective
    uid = (C.c_uint8 * 128)()
    L.check(klib.klara_comm_unique_id(uid), "comm_unique_id")
    comm = C.c_void_p()
    L.check(klib.klara_comm_init(C.byref(comm), 1, 0, uid, 0), "comm_init")
    for _ in range(2):
        cmean, cm2, cns, cna, cnt, cnc = ref.pooled_moments(comm)
        assert (cns, cna, cnt, cnc) == (ns, na, nt, nc)
        assert np.allclose(cmean, mean, rtol=1e-15, atol=0) and np.allclose(cm2, m2, rtol=1e-15, atol=0)
    L.check(klib.klara_comm_destroy(comm), "comm_destroy")
    ref.close()
    # two shards of the same job on this device, merged over gloo
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    out_q = ctx.Queue()
    procs = [ctx.Process(target=_gloo_rats_rank_main, args=(r, 2, port, out_q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [out_q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
    errs = [r[-1] for r in results if r[-1] is not None]
    assert not errs, errs
    for rank, offset, n, out, rs, rq, rnsaved, _ in results:
        assert np.array_equal(rs, s[offset:offset + n]) and np.array_equal(rq, q[offset:offset + n]) and rnsaved == nsaved     # sharding changes no bit
        assert out["nsamples"] == ns and out["naccept"] == na
        assert np.allclose(out["mean"], emean, rtol=1e-14, atol=1e-16)
        assert np.all(np.abs(out["var"] - evar) <= 1e-12 * evar), (rank, np.max(np.abs(out["var"] - evar) / evar))


def test_bench_two_ranks_on_one_device_logic():
    """`bench.py --gpus 2` the way the driver launches it (python -m torch.distributed.run, one process per rank), on ONE device with the
    gloo backend (`--same-device`: RCCL refuses two ranks on one GPU) — the rank / shard / barrier / max-over-ranks / summary all-reduce
    logic of the multi-GPU path, runnable on a one-GPU box: rank 0 prints one JSON line for both ranks' chains."""
    import json
    import socket
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           str(root / "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--reps", "3", "--backend", "gloo", "--same-device",
           "--no-extra", "--no-cpu-baseline", "--clock-warmup", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=str(root))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["warmup"] == 5 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["nchains_total"] == 2 * d["config"]["nchains_per_gpu"] and d["config"]["rccl_ranks_seen"] == 2
    assert d["value"] == pytest.approx(d["config"]["nchains_total"] * 20 / (d["ms_per_step"] * 20e-3), rel=1e-9)
    assert d["roofline"]["bound"] == "valu" and "cpu_baseline" not in d
    assert len(d["config"]["per_rank_ms_per_step"]) == 2 and d["config"]["rank_time_max_over_min"] >= 1.0
    assert max(d["config"]["per_rank_ms_per_step"]) == pytest.approx(d["ms_per_step"], rel=0.2)
    # two ranks on ONE device cannot share an RCCL communicator: the line says so and names the collective that ran instead
    col = d["config"]["collective"]
    assert col["requested"] == "klara" and "same-device" in col["error"] and col["used"].startswith("torch.distributed") and col["comm_nranks_rank_device"] is None
    assert col["ranks_by_transition_count"] == pytest.approx(2.0)


def test_bench_starts_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 2` with WORLD_SIZE unset — the form the driver uses for N = 1 — must not exit with a usage
    message: it starts its ranks itself (torch.distributed.run, 127.0.0.1) and rank 0's JSON line comes back through it."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--reps", "2", "--backend", "gloo",
                        "--same-device", "--no-extra", "--no-cpu-baseline", "--clock-warmup", "0"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=str(root))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["rccl_ranks_seen"] == 2 and d["value"] > 0


def test_bench_single_gpu_line_at_the_drivers_flags():
    """`python bench.py --gpus 1 --steps 20 --warmup 5` as the driver runs it (extras and CPU baseline off to keep the test short): one JSON
    line, the contract's keys, value = chains x steps / time, a VALU roofline below 1 whose launch duration was measured in this run."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--no-extra", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, cwd=str(root))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert (d["n_gpus"], d["steps"], d["warmup"], d["higher_is_better"], d["vs_baseline"], d["dtype"]) == (1, 20, 5, True, None, "f64")
    assert d["value"] == pytest.approx(65536 * 20 / (d["ms_per_step"] * 20e-3), rel=1e-9) and d["value"] > 1e9
    assert d["config"]["steps_per_launch"] == 32 and d["config"]["save_rule"].startswith("running sums")
    rf = d["roofline"]
    assert rf["bound"] == "valu" and 0.5 < rf["frac"] <= 1.0 and 300 < rf["launch_us"] < 700
    # frac is algorithmic: the instruction budget of the kernel that ran x 4 issue cycles / (launch duration x 1024 SIMDs x 2.4 GHz)
    assert rf["frac"] == pytest.approx(4.0 * rf["necessary_valu_insts_per_launch"] / (rf["launch_us"] * 1e-6) / (1024 * 2.4e9), rel=1e-9)
    assert rf["pmc"]["stale"] in (None, False, True) and (rf["utilisation"] is None or rf["frac"] <= rf["utilisation"] <= 1.0)


def test_bench_gathers_through_the_librarys_own_communicator():
    """bench.py's one exchange goes through the C ABI — klara_comm_unique_id -> (broadcast) -> klara_comm_init ->
    klara_gather_moments(h, comm, ...) — the path a Julia binding calls (/root/reference/src/jobs/jobs.jl:212 is what it replaces), not
    through torch.distributed.  On a one-GPU box: `--force-comm` makes the one-rank RCCL communicator, so every call of the wiring runs;
    `--collective both` gathers a second time through the Python mirror and the line carries both timings and their agreement."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--reps", "3", "--force-comm", "--collective", "both",
                        "--no-extra", "--no-cpu-baseline", "--clock-warmup", "0"], capture_output=True, text=True, timeout=600, env=env, cwd=str(root))
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith('{"metric"')][0])
    col = d["config"]["collective"]
    assert col["error"] is None and "klara_gather_moments" in col["used"] and col["comm_nranks_rank_device"] == [1, 0, 0]
    assert d["config"]["rccl_ranks_seen"] == 1 and col["ranks_by_transition_count"] == pytest.approx(1.0)
    assert "C ABI" in d["config"]["parallelism"] and d["config"]["summary_gather_ms"] > 0
    ab = col["ab"]
    assert ab["same_nsamples"] and ab["same_naccept"] and ab["max_rel_mean_var_difference"] < 1e-12 and ab["klara_ms"] > 0 and ab["torch_ms"] > 0
    assert 0.0 < d["config"]["acceptance_rate"] < 1.0
    # --scaling strong: the fixed-size job (65,536 chains whatever the rank count) — at one rank the same job, named as such
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--reps", "2", "--scaling", "strong",
                        "--force-comm", "--no-extra", "--no-cpu-baseline", "--clock-warmup", "0"], capture_output=True, text=True, timeout=600, env=env, cwd=str(root))
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith('{"metric"')][0])
    assert d["scaling"] == "strong" and d["config"]["nchains_total"] == 65536 and d["config"]["rccl_ranks_seen"] == 1
