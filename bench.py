#!/usr/bin/env python3
"""bench.py — whole-job MCMC transition throughput on MI355X (BASELINE.json metric).

Workload at N=1 (BASELINE.json configs[1]): MALA driftstep=0.9, lt = -|x|^2 on D=100, 65,536 chains,
x0 ~ N(0, I) from the Philox init stream, VanillaMCTuner, state resident in HBM before the timed region.
A "step" is one transition (one `iterate!`) of every chain.  With --gpus N every rank owns its own
65,536-chain shard (weak scaling, global chain ids = rank*65536 + local), no data-path collective; the
only exchange is the end-of-run all-reduce of pooled chain summaries over RCCL, inside the timed region.

Prints ONE JSON line (rank 0).  Extra keys beyond the driver's contract:
  roofline      dominant transition kernel (k_diagt<MALA>, layout kind 3): algorithmic HBM bytes per launch / mean launch
                duration from HIP events on the launch stream, measured in a separate pass with every launch on one stream
                (the timed region overlaps two half-size launches on two streams: config.streams); DESIGN.md sections 4-5
                give the per-unit figures; `traffic` = PMC bytes of the same command (profiles/)
  cpu_baseline  the CPU oracle ("port" of the reference path) timed on this box's host cores on a bounded
                sample of the same workload (N=1 only)
  extra         secondary measurements outside the timed region: fused launches, HMC leapfrog rates (diagonal and dense /
                FP64-MFMA target), and BASELINE configs 4 and 5 at their per-GPU share
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

NCHAINS_PER_GPU = 65536
NDIMS = 100
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
FP64_MFMA_PEAK_TF = 78.6    # FP64 matrix = FP64 vector peak on MI355X


def algorithmic_bytes_per_launch(nchains, d, sampler, spl, summaries, accept_rate=1.0):
    """carried state S = 2*D*8+8 (x, g, lt) for MALA/HMC, D*8+8 for MH/Slice.  A launch reads it once;
    it has to write it back only for the chains that moved (a rejected proposal leaves x, g, lt as they were), so the
    write side is scaled by the measured fraction of chains that accepted at least once in the launch (for one
    transition per launch: the acceptance rate; SURVEY's 2*S+1 figure is the accept_rate = 1 upper end).
    The per-chain accept counter (8 B RMW, accepted chains only) and, when on, the running sums (2 arrays RMW) count too."""
    s = (2 * d * 8 + 8) if sampler in ("mala", "hmc") else (d * 8 + 8)
    b = s + accept_rate * (s + 16)
    if summaries:
        b += 4 * d * 8
    return nchains * b


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--spl", type=int, default=1, help="transitions fused per kernel launch (1 = one iterate! per launch)")
    ap.add_argument("--chains", type=int, default=NCHAINS_PER_GPU, help="chains per GPU")
    ap.add_argument("--streams", type=int, default=0,
                    help="internal streams for independent chain partitions (0 = library default, 1 = every launch on the caller's stream)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for logic tests)")
    ap.add_argument("--same-device", action="store_true",
                    help="logic test only: every rank uses cuda:0 (needs --backend gloo; RCCL refuses duplicate GPUs)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import klara_jl_amd as K
    from klara_jl_amd import _lib as L

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the transition path has no CPU fallback)")
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    cdev = "cuda" if args.backend == "nccl" else "cpu"     # where the (tiny) collectives' tensors live

    n = args.chains
    total_steps = args.warmup + args.steps
    stream = torch.cuda.current_stream().cuda_stream
    eng = K.Engine(sampler=L.SAMPLER_MALA, target=K.GaussDiagTarget.negdot(NDIMS), nchains=n, nsteps=total_steps,
                   burnin=0, driftstep=0.9, seed=20260927, chain_offset=rank * n, device=local_rank,
                   monitor=0, steps_per_launch=args.spl, stream=stream, nstreams=args.streams)
    eng.init_state_normal()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    eng.run(args.warmup)
    if dist is not None:   # warm the communicator outside the timed region
        t = torch.zeros(4, device=cdev); dist.all_reduce(t)
    barrier()
    t0 = time.perf_counter()
    eng.run(args.steps)
    if dist is not None:
        summ = K.gather_engine_summaries(eng)          # RCCL all-reduce of chain summaries only
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if dist is not None:
        te = torch.tensor([elapsed], device=cdev, dtype=torch.float64)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    kernel_ms, nlaunch = eng.last_run_ms()
    lay_kind, lay_g, lay_e = eng.layout()
    _, _, nacc, ntr, _ = eng.pooled_summaries(with_sums=False)
    acc_rate = nacc / max(ntr, 1)

    out = None
    if rank == 0:
        transitions = float(n) * world * args.steps
        value = transitions / elapsed
        out = {
            "metric": "MCMC transitions/sec (whole node), 100-dim Gaussian, 65k chains",
            "value": value, "unit": "transitions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: MALA driftstep=0.9, lt=-|x|^2, D=100, 65,536 chains per GPU, "
                                   "VanillaMCTuner, x0~N(0,I)",
                       "nchains_per_gpu": n, "ndims": NDIMS, "steps_per_launch": args.spl,
                       "parallelism": f"chains sharded over {world} GPU(s), no data-path collective",
                       "streams": "library default (2 chain partitions on 2 HIP streams)" if args.streams == 0 else args.streams,
                       "acceptance_rate": acc_rate,
                       "timed_region_kernel_ms_per_step": kernel_ms / max(nlaunch, 1) * (1 if args.spl <= 1 else 1.0 / args.spl)},
        }
    eng.close()

    if rank == 0:
        out["roofline"] = roofline_pass(K, L, n, args.spl, rank, local_rank, stream, acc_rate)
    if rank == 0 and world == 1 and not args.no_extra:
        out["extra"] = extra_measurements(K, L, n, stream)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(L)
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def roofline_pass(K, L, n, spl, rank, local_rank, stream, acc_rate):
    """Duration of the dominant kernel, one launch at a time: the same workload with every launch on the caller's
    stream (nstreams=1; the timed region above overlaps two half-size launches on two streams, which says nothing
    about a single launch), HIP events around 200 launches on that stream.  achieved = algorithmic bytes / duration."""
    e = K.Engine(sampler=L.SAMPLER_MALA, target=K.GaussDiagTarget.negdot(NDIMS), nchains=n, nsteps=100000, burnin=0,
                 driftstep=0.9, seed=20260927, chain_offset=rank * n, device=local_rank, monitor=0, steps_per_launch=spl,
                 stream=stream, nstreams=1)
    e.init_state_normal()
    e.run(40 * max(spl, 1)); e.run(200 * max(spl, 1))
    kernel_ms, nlaunch = e.last_run_ms()
    lay_kind, lay_g, lay_e = e.layout()
    e.close()
    launch_s = kernel_ms * 1e-3 / max(nlaunch, 1)
    alg = algorithmic_bytes_per_launch(n, NDIMS, "mala", spl, False, acc_rate if spl <= 1 else 1.0)
    achieved = alg / launch_s / 1e9
    rf = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
          "traffic": None,
          "kernel": (f"k_diagt<MALA, NP={lay_e // 2}, Q={lay_g}> (pair-transposed layout: {lay_g} lanes x {lay_e // 2} element "
                     f"pairs per chain, {64 // lay_g} chains per wavefront)") if lay_kind == 3 else
                    (f"k_transitions<MALA, GAUSS_DIAG, E={lay_e}> ({lay_g} lanes x {lay_e} elements per chain, "
                     f"{64 // lay_g} chains per wavefront)"),
          "algorithmic_bytes_per_launch": alg, "launch_us": launch_s * 1e6, "launches": nlaunch,
          "chains_per_launch": n, "accept_rate_used": acc_rate if spl <= 1 else 1.0,
          "survey_2S_plus_1_bytes_per_launch": n * (2 * (2 * NDIMS * 8 + 8) + 1),
          "note": ("the kernel is bound by FP64/INT VALU issue (in-kernel Philox + Box-Muller), not by HBM: see DESIGN.md "
                   "section 5; measured with nstreams=1, one launch at a time.  The algorithmic bytes are the algorithm's "
                   "(x, gradient and log-target read per chain); the kernel re-forms the gradient from x instead of reading "
                   "it, so the measured traffic is about half of them")}
    # HBM bytes per launch from the PMC passes of the same workload (profiles/, scripts/profile_bench.sh)
    try:
        tr = json.loads((ROOT / "profiles" / "r1_bench_mala_traffic.json").read_text())
        kname = f"k_diagt<1, {lay_e // 2}, {lay_g}," if lay_kind == 3 else f"k_transitions<1, 0, {lay_e},"
        if (tr["nchains"], tr["ndims"], tr["steps_per_launch"]) == (n, NDIMS, spl) and kname in tr["kernel"]:
            rf["traffic"] = tr["traffic_bytes_per_launch"]
            rf["traffic_gbs"] = tr["traffic_bytes_per_launch"] / launch_s / 1e9
            rf["traffic_source"] = "profiles/r1_bench_mala_traffic.json (FETCH_SIZE x2 + WRITE_SIZE, KiB)"
    except Exception:
        pass
    return rf


def extra_measurements(K, L, n, stream):
    """Outside the timed region: fused launches of the same workload, and the north-star HMC rates."""
    import numpy as np
    ex = {}
    for spl in (16,):
        e = K.Engine(sampler=L.SAMPLER_MALA, target=K.GaussDiagTarget.negdot(NDIMS), nchains=n, nsteps=100000,
                     driftstep=0.9, steps_per_launch=spl, stream=stream)
        e.init_state_normal(); e.run(64)
        t0 = time.perf_counter(); e.run(512); dt = time.perf_counter() - t0
        ex[f"mala_iso_spl{spl}_transitions_per_s"] = n * 512 / dt
        e.close()
    # HMC L=10 eps=0.1 on the README target (HBM/VALU-bound) and on the dense target (FP64-MFMA-bound; cfg 3)
    for key, target in (("hmc_iso", K.GaussDiagTarget.negdot(NDIMS)),
                        ("hmc_dense", K.GaussDenseTarget.compound_symmetric(NDIMS, 0.5))):
        e = K.Engine(sampler=L.SAMPLER_HMC, target=target, nchains=n, nsteps=100000, leapstep=0.1, nleaps=10,
                     steps_per_launch=16, stream=stream)      # 16 = the library's default fusion
        e.init_state_normal(); e.run(16)
        steps = 128
        t0 = time.perf_counter(); e.run(steps); dt = time.perf_counter() - t0
        ms, nl = e.last_run_ms()
        ex[f"{key}_leapfrog_chain_per_s"] = n * steps * 10 / dt
        ex[f"{key}_transitions_per_s"] = n * steps / dt
        if key == "hmc_dense":
            flops = n * steps * 10 * (2 * NDIMS * NDIMS + 6 * NDIMS)
            ex["hmc_dense_fp64_tflops"] = flops / (ms * 1e-3) / 1e12
            ex["hmc_dense_frac_of_fp64_mfma_peak"] = ex["hmc_dense_fp64_tflops"] / FP64_MFMA_PEAK_TF
        e.close()
    # the two data-model configurations of BASELINE.json at their per-GPU share (cfg 4: 262,144 / 8 chains of the swiss
    # logistic regression, MALA h = 0.1; cfg 5: 1,048,576 / 8 chains of the rats hierarchical model, HMC L = 32 with the
    # per-GPU pooled AcceptanceRate tuner).  Data: the reference's own files as committed fixtures (tests/golden/*.npz).
    try:
        gold = ROOT / "tests" / "golden"
        sw = np.load(gold / "swiss.npz")
        X = sw["measurements"]; X = np.ascontiguousarray((X - X.mean(axis=0)) / X.std(axis=0, ddof=1))
        y = np.ascontiguousarray(sw["status"].astype(np.float64))
        nc = 32768
        x0 = np.array([5.1, -0.9, 8.2, -4.5])[None, :] + 0.1 * np.random.default_rng(0).standard_normal((nc, 4))
        e = K.Engine(sampler=L.SAMPLER_MALA, target=K.LogisticTarget(X, y, 100.0), nchains=nc, nsteps=10 ** 6, driftstep=0.1,
                     steps_per_launch=50, monitor=0, stream=stream)
        e.set_state(x0); e.run(100)
        t0 = time.perf_counter(); e.run(500); dt = time.perf_counter() - t0
        ex["cfg4_swiss_logistic_mala_transitions_per_s_per_gpu"] = nc * 500 / dt
        e.close()
        rats = np.load(gold / "rats.npz")
        t = K.HierNormalTarget(rats["weight"], rats["age"] - 22.0)
        nc = 131072
        x0 = t.least_squares_start()[None, :] + 0.05 * np.random.default_rng(1).standard_normal((nc, t.ndims))
        e = K.Engine(sampler=L.SAMPLER_HMC, target=t, nchains=nc, nsteps=2000, burnin=1000, leapstep=0.02, nleaps=32,
                     tuner=L.TUNER_ACCEPT_RATE, tuner_mode=L.TUNE_POOLED, targetrate=0.65, period=100, steps_per_launch=10,
                     monitor=L.MON_SUMMARIES, stream=stream)
        e.set_state(x0); e.run(100)
        t0 = time.perf_counter(); e.run(200); dt = time.perf_counter() - t0
        ex["cfg5_rats_hmc_L32_leapfrog_chain_per_s_per_gpu"] = nc * 200 * 32 / dt
        ex["cfg5_rats_hmc_L32_transitions_per_s_per_gpu"] = nc * 200 / dt
        e.close()
    except Exception as exc:      # the fixtures are part of the repository; a failure here must not lose the headline line
        ex["model_configs_error"] = repr(exc)
    return ex


def cpu_baseline(L):
    """CPU oracle (restatement of the reference path, OpenMP over chains) on a bounded sample of the workload."""
    import numpy as np
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle_ffi as O
    cores = usable_cores()
    lib = O.load()
    lib.ko_set_num_threads(int(cores))
    nch, chunk = 32 * cores, 25
    job = O.OracleJob(sampler=L.SAMPLER_MALA, target_kind=L.TARGET_GAUSS_DIAG, nchains=nch, ndims=NDIMS,
                      nsteps=10 ** 9, driftstep=0.9, want_accept=False, want_sums=False)
    job.init_state_normal()
    job.run(chunk)                                   # warm-up (thread pool, caches)
    # bounded sample: fixed-size chunks until ~12 s of wall time have been spent
    steps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < 12.0:
        job.run(chunk); steps += chunk
    dt = time.perf_counter() - t0
    out = {"value": nch * steps / dt, "unit": "transitions/s", "cores": cores, "kind": "port",
           "sample": f"oracle/libklara_oracle.so (C restatement, gcc -O2, OpenMP over chains, {cores} threads): "
                     f"MALA driftstep=0.9, D=100, {nch} chains x {steps} transitions in {dt:.1f} s"}
    # the reference's own execution model is one chain after another on one thread (BasicMCJob.jl:212-244): 3 s sample
    lib.ko_set_num_threads(1)
    one = O.OracleJob(sampler=L.SAMPLER_MALA, target_kind=L.TARGET_GAUSS_DIAG, nchains=32, ndims=NDIMS, nsteps=10 ** 9,
                      driftstep=0.9, want_accept=False, want_sums=False)
    one.init_state_normal(); one.run(chunk)
    steps1, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < 3.0:
        one.run(chunk); steps1 += chunk
    out["single_thread_value"] = 32 * steps1 / (time.perf_counter() - t0)
    return out


def usable_cores():
    """Host threads this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


if __name__ == "__main__":
    main()
