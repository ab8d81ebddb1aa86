#!/usr/bin/env python3
"""bench.py — whole-job MCMC transition throughput on MI355X (BASELINE.json metric).

Workload at N = 1 (BASELINE.json configs[1]): MALA driftstep = 0.9, lt = -|x|^2 on D = 100, 65,536 chains, x0 ~ N(0, I) from the
Philox init stream, VanillaMCTuner, state resident in HBM before the timed region.  A "step" is one transition (one `iterate!`,
src/samplers/iterate/MALA.jl:78-153) of every chain INCLUDING the save rule of `run(job)` (BasicMCJob.jl:226-231): every step is
in the post-burn-in range and is accumulated into the per-chain running sums (KLARA_MON_SUMMARIES) that `mean(chain)` reads.
`klara_run` IS the `for i in 1:nsteps` loop (BasicMCJob.jl:219-238): the library's default of 32 transitions per kernel launch is
the headline (config.steps_per_launch), one transition per launch is reported in `extra`.

Timed region: `--reps` (5) repetitions of exactly `--steps` transitions, each bracketed by a barrier + torch.cuda.synchronize()
on both sides; a rank's time runs from leaving the opening barrier to its own synchronize() after the K steps, the repetition's
time is the MAX over ranks (the closing barrier's own latency is not part of any rank's K steps); `value` uses the MEDIAN repetition.  With --gpus N every rank owns its own shard
(weak scaling: 65,536 chains per GPU, global chain ids = rank * 65,536 + local; `--scaling strong`: `--total-chains` sharded over
the ranks), no data-path collective; the only exchange is the end-of-run reduction of the chain summaries — pooled on the device
and, for N > 1, all-reduced over RCCL THROUGH THE LIBRARY'S OWN C ABI (`--collective klara`, the default: rank 0 calls
klara_comm_unique_id, the 128 bytes go round over the launcher's rendezvous (torch.distributed, gloo: bootstrap, barriers and the
max-over-ranks of the times only), every rank calls klara_comm_init and the exchange is klara_gather_moments(h, comm, ...): the
path a Julia binding calls, include/klara_hip.h) — once, after the job's last transition; it is timed on its own
(config.summary_gather_ms: an end-of-run cost of ~0.1 ms does not belong inside a timed region that the driver may make 20
transitions short).  `--collective torch` gathers through the Python mirror instead (klara.jl_amd/distributed.py: device ->
NumPy -> torch.distributed.all_reduce over `--torch-backend`), `--collective both` times the two side by side.

Prints ONE JSON line (rank 0).  Keys beyond the driver's contract:
  roofline      the dominant transition kernel.  `bound` = "valu": the kernel is bound by vector-ALU issue (in-kernel Philox +
                Box-Muller), HBM moves a few per cent of its peak (`hbm`).  `frac` is ALGORITHMIC: `achieved` = the vector
                instructions the algorithm needs per launch (scripts/instruction_budget.py: one count per operation of the stream
                functions and of the sampler arithmetic, derivation in profiles/README.md) x 4 issue cycles / the mean launch
                duration measured HERE with HIP events on the launch stream (one launch at a time, nstreams = 1); `peak` = 1024
                SIMDs x 2.4 GHz.  `utilisation` is the VALU-busy fraction (SQ_ACTIVE_INST_VALU of the committed PMC summary
                profiles/r4_pmc_kernels.json / the same duration) and `issued_over_necessary` the instructions the kernel really
                issues (SQ_INSTS_VALU) over the budget; `pmc.stale` is true when the loaded kernel's registers / scratch differ
                from the ones the counters were collected on.  `hbm` gives the byte side: SURVEY section 8(d)'s contract bytes,
                the bytes this kernel has to move, the PMC traffic and the fraction of 8 TB/s each amounts to.
  cpu_baseline  the CPU oracle ("port" of the reference path, OpenMP over chains) on a bounded sample of the same workload
  extra         the other configurations, each with its own {bound, frac, kernel, source} object
"""
import argparse
import json
import os
import statistics
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

NCHAINS_PER_GPU = 65536
NDIMS = 100
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
FP64_MFMA_PEAK_TF = 78.6    # FP64 matrix = FP64 vector peak on MI355X
CLOCK_HZ = 2.4e9            # MI355X_MICROARCH.md: max clock; 256 CUs x 4 SIMDs
NSIMD = 1024
# How the running sums of a moving chain are kept (4-lane kernels + atomic folds, or 8-lane kernels + resident sums) is decided by the
# library itself, on the device, launch by launch (klara_desc.sparse_moves = 0): no caller hint.
PMC_JSON = ROOT / "profiles" / "r6_pmc_kernels.json"
PMC_EXPECT = {}             # filled by main(): the launch length the committed counters must have been collected at


def budgets():
    """scripts/instruction_budget.py: necessary vector instructions per wavefront and transition of the kernels priced here"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("instruction_budget", ROOT / "scripts" / "instruction_budget.py")
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    return mod.BUDGETS


def pmc_lookup(kernel_sub, grid=None):
    """Mean per-launch counters of the kernel whose name contains `kernel_sub` (and whose grid matches) from the committed PMC
    summary (scripts/profile_round.sh -> scripts/make_pmc_json.py).  None when the file has no such kernel."""
    try:
        doc = json.loads(PMC_JSON.read_text())
        rows = doc["kernels"]
        # the per-launch counters of the headline kernel only hold for the launch length they were collected at
        if "k_diagt<1" in kernel_sub and doc.get("bench_config", {}).get("steps_per_launch") not in (None, PMC_EXPECT.get("steps_per_launch")) \
                and "false, true, true" in kernel_sub:
            return None
    except Exception:
        return None
    best = None
    for r in rows:
        if kernel_sub in r["kernel"] and (grid is None or r.get("grid") == grid):
            if best is None or r["counters"].get("SQ_INSTS_VALU", {}).get("n", 0) > best["counters"].get("SQ_INSTS_VALU", {}).get("n", 0):
                best = r
    return best


def pmc_recorded_attributes():
    """{lookup key: [vgprs, scratch bytes]} of the kernels as they were loaded when the committed counters were collected"""
    try:
        return json.loads(PMC_JSON.read_text()).get("loaded_kernel_attributes", {})
    except Exception:
        return {}


def valu_roofline(kernel_sub, launch_s, grid=None, label=None, necessary_per_launch=None, attrs=None, budget=None):
    """{bound: valu, achieved, peak, frac, utilisation, ...}.  frac = necessary VALU issue-cycles per launch (instruction budget x 4)
    / (launch duration x 1024 SIMDs x 2.4 GHz) — None for a kernel without a budget; utilisation = VALU-busy SIMD-cycles per launch
    (PMC) / the same denominator.  attrs = (vgprs, scratch bytes, static LDS) of the kernel as loaded (klara_get_kernel_attributes):
    a PMC row collected on a kernel with other registers / scratch is marked stale and not used."""
    peak = NSIMD * CLOCK_HZ
    rf = {"bound": "valu", "achieved": None, "peak": peak, "unit": "necessary VALU issue-cycle/s", "frac": None, "traffic": None, "utilisation": None,
          "kernel": label or kernel_sub, "launch_us": launch_s * 1e6,
          "source": "frac: necessary vector instructions per launch (scripts/instruction_budget.py, profiles/README.md) x 4 issue cycles / launch "
                    "duration from HIP events in this run / (1024 SIMDs x 2.4 GHz); utilisation: SQ_ACTIVE_INST_VALU (quad-cycles x 4, chip "
                    f"total) per launch from profiles/{PMC_JSON.name} / the same denominator"}
    if necessary_per_launch is not None:
        rf.update(achieved=4.0 * necessary_per_launch / launch_s, frac=4.0 * necessary_per_launch / launch_s / peak,
                  necessary_valu_insts_per_launch=necessary_per_launch)
    if budget is not None:
        rf["budget"] = {k: v for k, v in budget.items()}
        if necessary_per_launch is not None and budget.get("extra_issue_units_per_wave_transition") is not None:
            # the same fraction with v_mad_u64_u32 (x 1.6) and v_rsq_f64 / v_rcp_f64 (x 3.2) at their measured issue cost
            # (profiles/r3_ubench_instruction_costs.txt) instead of one plain instruction each
            w = 1.0 + budget["extra_issue_units_per_wave_transition"] / budget["per_wave_transition"]
            rf["frac_at_measured_instruction_costs"] = rf["frac"] * w
    row = pmc_lookup(kernel_sub, grid)
    pmc = {"file": f"profiles/{PMC_JSON.name}", "key": kernel_sub, "kernel": None, "stale": None}
    rf["pmc"] = pmc
    if attrs is not None:       # (vgprs rounded up to the allocation granule of 8, scratch bytes: what scripts/make_pmc_json.py records per key)
        pmc.update(loaded_vgpr=8 * ((attrs[0] + 7) // 8), loaded_scratch=attrs[1])
    if row is None or "SQ_ACTIVE_INST_VALU" not in row["counters"]:
        pmc["why"] = "no row for this kernel / grid in the committed summary"
        return rf
    pmc["kernel"] = row["kernel"]
    rec = pmc_recorded_attributes().get(kernel_sub)
    if attrs is not None and rec is not None:
        pmc.update(recorded_vgpr=rec[0], recorded_scratch=rec[1], stale=bool([pmc["loaded_vgpr"], pmc["loaded_scratch"]] != list(rec)))
        if pmc["stale"]:
            pmc["why"] = "the loaded kernel's registers / scratch differ from the build the counters were collected on: counters not used"
            return rf
    c = row["counters"]
    busy = 4.0 * c["SQ_ACTIVE_INST_VALU"]["mean"]
    rf.update(utilisation=busy / launch_s / peak, valu_busy_cycles_per_simd=busy / NSIMD,
              issued_valu_insts_per_launch=c.get("SQ_INSTS_VALU", {}).get("mean"), waves_per_launch=c.get("SQ_WAVES", {}).get("mean"))
    if necessary_per_launch is not None and rf["issued_valu_insts_per_launch"]:
        rf["issued_over_necessary"] = rf["issued_valu_insts_per_launch"] / necessary_per_launch
    if "GRBM_GUI_ACTIVE" in c:      # (chip clock during the PMC pass: the sum over the 8 XCDs / 8 / the launch duration of that pass is not known here)
        rf["pmc_gui_active_cycles_per_xcd"] = c["GRBM_GUI_ACTIVE"]["mean"] / 8.0
    return rf


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1024)
    ap.add_argument("--warmup", type=int, default=128)
    ap.add_argument("--reps", type=int, default=5, help="repetitions of the timed region; the median is reported")
    ap.add_argument("--spl", type=int, default=0, help="transitions per kernel launch (0 = library default 32; 1 = one iterate! per launch)")
    ap.add_argument("--chains", type=int, default=NCHAINS_PER_GPU, help="chains per GPU (weak scaling)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--total-chains", type=int, default=NCHAINS_PER_GPU, help="--scaling strong: chains of the whole job, sharded over the ranks")
    ap.add_argument("--streams", type=int, default=0,
                    help="internal streams for independent chain partitions (0 = library default, 1 = every launch on the caller's stream)")
    ap.add_argument("--clock-warmup", type=int, default=3200,
                    help="transitions of an identical scratch job run straight before the timed repetitions so that the device clocks "
                         "are at their steady state (0 = off); the measured job gets exactly --warmup steps")
    ap.add_argument("--no-save", action="store_true", help="drop the save rule (no running sums): the transition kernel alone")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--collective", choices=("klara", "torch", "both"), default="klara",
                    help="the job's one exchange (summary gather): klara = the library's own RCCL communicator behind the C ABI "
                         "(klara_comm_init / klara_gather_moments); torch = the Python mirror over torch.distributed; both = klara, then torch for the A/B")
    ap.add_argument("--backend", default="gloo",
                    help="torch.distributed backend of the launcher's rendezvous: id broadcast, barriers, max-over-ranks of the times "
                         "(gloo: host-side only, the GPUs' one communicator is the library's; nccl = a second RCCL communicator made by torch)")
    ap.add_argument("--torch-backend", default="nccl", help="--collective torch / both: backend of the group the Python mirror all-reduces over")
    ap.add_argument("--comm-timeout", type=float, default=60.0, help="seconds klara_comm_init may take before this rank gives up on the library's communicator")
    ap.add_argument("--force-comm", action="store_true",
                    help="make the library's communicator even for one rank (a one-rank RCCL communicator: the wiring, testable on a one-GPU box)")
    ap.add_argument("--same-device", action="store_true",
                    help="logic test only: every rank uses cuda:0 (RCCL refuses duplicate GPUs: the gather falls back to --collective torch over gloo)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves, exactly as the driver would
        # (one process per GPU, rendezvous on 127.0.0.1), hand their output through and leave with their exit code
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        print("bench.py: no launcher (WORLD_SIZE unset): starting " + " ".join(cmd[1:7]), file=sys.stderr, flush=True)
        raise SystemExit(subprocess.call(cmd))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} was started with WORLD_SIZE={world}: launch it with --nproc-per-node {args.gpus} "
                         "(or without a launcher: it starts its own ranks)")

    import numpy as np
    import torch
    import klara_jl_amd as K
    from klara_jl_amd import _lib as L

    if not torch.cuda.is_available():
        raise SystemExit(f"bench.py needs a GPU (the transition path has no CPU fallback) [rank {rank} of {world}]")
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
    cdev = "cuda" if args.backend == "nccl" else "cpu"     # where the rendezvous group's (tiny) tensors live
    # -- the job's one collective.  klara: the library's communicator over all ranks, made through the C ABI alone; the rendezvous group
    # carries the 128-byte id and nothing else.  A failure to make it is reported in the line (config.collective_error) and the gather
    # falls back to the Python mirror: the timed region has no collective in it, so `value` does not depend on which one ran.
    collective = args.collective
    comm, comm_info, collective_error = None, None, None
    if args.same_device and world > 1 and collective != "torch":
        collective, collective_error = "torch", "--same-device: RCCL refuses two ranks on one GPU; gathered over the rendezvous group instead"
    comm_abandoned = False
    if collective in ("klara", "both") and (world > 1 or args.force_comm):
        # (one node: RCCL's bootstrap sockets over the loopback interface — the ranks are in one container whose other interfaces / host name need not be
        # reachable from inside it; the data path is xGMI / shared memory either way)
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # klara_comm_init returns when every rank has joined; a rank that cannot reach rank 0's bootstrap address would wait in there for as long as RCCL
        # retries.  bootstrap_comm runs that one call on a helper thread with a deadline: past it this rank reports the failure, every rank falls back to
        # the Python mirror together (the MIN all-reduce below), and the line still comes out — without the product's collective in it, and saying so.
        try:
            bc = K.torch_broadcast_bytes() if dist is not None else (lambda b: b)
            comm = K.bootstrap_comm(L.load(), rank, world, local_rank, bc, timeout=args.comm_timeout)
            comm_info = comm.info()
        except K.CommBootstrapTimeout as exc:
            comm, comm_abandoned, collective_error = None, True, repr(exc)
        except Exception as exc:
            comm, collective_error = None, repr(exc)
        if dist is not None:        # every rank has a communicator, or none uses it (a rank on its own in a collective would hang)
            ok = torch.tensor([1 if comm is not None else 0], device=cdev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                if comm is not None:
                    comm = None         # (not destroyed: a peer of this communicator never joined or failed — ncclCommDestroy could wait for it)
                collective_error = collective_error or "another rank failed to make the communicator"
                collective = "torch"
        elif comm is None:
            collective = "torch"
    tgroup, tgroup_error = None, None
    if dist is not None and args.collective in ("torch", "both") and collective_error is None and not args.same_device and args.torch_backend != args.backend:
        try:                                        # the mirror's own group (nccl = RCCL made by torch) for the A/B
            tgroup = dist.new_group(backend=args.torch_backend)
        except Exception as exc:
            tgroup_error = repr(exc)

    if args.scaling == "strong":
        offset, n = K.shard_chains(args.total_chains, rank, world)
        n_total = args.total_chains
    else:
        n, offset, n_total = args.chains, rank * args.chains, args.chains * world
    spl = args.spl if args.spl > 0 else L.DEFAULT_STEPS_PER_LAUNCH
    PMC_EXPECT["steps_per_launch"] = spl
    monitor = 0 if args.no_save else L.MON_SUMMARIES
    total_steps = args.warmup + args.reps * args.steps
    stream = torch.cuda.current_stream().cuda_stream
    eng = K.Engine(sampler=L.SAMPLER_MALA, target=K.GaussDiagTarget.negdot(NDIMS), nchains=n, nsteps=total_steps,
                   burnin=0, driftstep=0.9, seed=20260927, chain_offset=offset, device=local_rank,
                   monitor=monitor, steps_per_launch=spl, stream=stream, nstreams=args.streams)
    eng.init_state_normal()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def gather(which):
        """the job's one exchange -> summaries over every rank's chains"""
        if which == "klara" and comm is not None:
            return K.gather_engine_moments_klara(eng, comm)
        return K.gather_engine_summaries(eng, group=tgroup)

    eng.run(args.warmup)
    if dist is not None:   # warm the communicators outside the timed region
        t = torch.zeros(4, device=cdev); dist.all_reduce(t)
    gather("klara" if comm is not None else "torch")
    if collective == "both":
        gather("torch")
    # Device warm-up.  An MI355X that has been idle for a few milliseconds (job creation is enough) runs the first ~20 ms of
    # work at reduced clocks: the same 20-transition launches take 21.5 us per transition on a cold device and 17.5 us when
    # it has just been busy (scripts/probe_short_region.py, profiles/r2_short_region_probe.txt).  A timed region of
    # `--steps 20` lasts 0.4 ms and would measure the clock ramp, not the kernel, so the device is kept busy for ~50 ms on an
    # identical SCRATCH job straight before the timed repetitions; the measured job itself gets exactly `--warmup` steps.
    if args.clock_warmup > 0:
        scratch = K.Engine(sampler=L.SAMPLER_MALA, target=K.GaussDiagTarget.negdot(NDIMS), nchains=n, nsteps=10 ** 7, burnin=0,
                           driftstep=0.9, seed=1, chain_offset=offset, device=local_rank, monitor=monitor, steps_per_launch=spl,
                           stream=stream, nstreams=args.streams)
        scratch.init_state_normal()
        scratch.run(args.clock_warmup)          # (closed after the timed repetitions: freeing memory would idle the device again)
    times, kernel_ms_per_step, summ, rank_times = [], [], None, []
    for _ in range(args.reps):
        barrier()
        t0 = time.perf_counter()
        eng.run_async(args.steps)                   # klara_run_async: enqueue the K steps (the launch loop of klara_run) ...
        torch.cuda.synchronize()                    # ... and the contract's synchronisation: the device has finished them
        elapsed = time.perf_counter() - t0          # this rank's K steps; the job's time is the MAX over ranks (below)
        eng.synchronize()                           # klara_synchronize (outside the clock; nothing left to wait for): raises what a kernel flagged
        if dist is not None:
            dist.barrier()                           # closing bracket: every rank is done before anyone goes on
            te = torch.tensor([elapsed], device=cdev, dtype=torch.float64)
            every = [torch.zeros_like(te) for _ in range(world)]
            dist.all_gather(every, te)               # every rank's own K steps: the job's time is their MAX; the spread shows imbalance
            rank_times.append([float(t.item()) for t in every])
            elapsed = max(rank_times[-1])
        else:
            rank_times.append([elapsed])
        times.append(elapsed)
        kms, nl = eng.last_run_ms()
        kernel_ms_per_step.append(kms / args.steps)
    elapsed = statistics.median(times)
    if args.clock_warmup > 0:
        scratch.close()
    # the job's one exchange, after its last transition: chain summaries pooled on the device (+ RCCL all-reduce for N > 1)
    def timed_gather(which):
        barrier()
        t0 = time.perf_counter()
        out_ = gather(which)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        return out_, (time.perf_counter() - t0) * 1e3
    primary = "klara" if comm is not None else "torch"
    summ, gather_ms = timed_gather(primary)
    gather_ab = None
    if collective == "both" and comm is not None:
        summ_t, ms_t = timed_gather("torch")
        gather_ab = {"klara_ms": gather_ms, "torch_ms": ms_t,
                     "torch_group": (args.torch_backend if tgroup is not None else args.backend) if dist is not None else "no process group (one rank)",
                     "same_nsamples": bool(summ_t["nsamples"] == summ["nsamples"]), "same_naccept": bool(summ_t["naccept"] == summ["naccept"]),
                     "max_rel_mean_var_difference": (float(max(np.max(np.abs(summ_t["mean"] - summ["mean"]) / (np.abs(summ["mean"]) + 1e-300)),
                                                               np.max(np.abs(summ_t["var"] - summ["var"]) / summ["var"])))
                                                     if "var" in summ and "var" in summ_t else None)}
    lay_kind, lay_g, lay_e = eng.layout()
    launch_counts = eng.launch_modes()[0] if hasattr(L.load(), 'klara_get_launch_modes') else [0, 0, 0]
    acc_rate = float(summ["acceptance"]) if summ is not None and "acceptance" in summ else None
    # ranks that took part in the gather: from the communicator itself (ncclCommCount) when the library's collective ran, and — either
    # way — from what came back (every rank contributes n chains' transitions; --scaling strong: the job's chains)
    ranks_by_count = (float(summ["ntransitions"]) / max(1, (args.warmup + args.reps * args.steps)) / (n if args.scaling == "weak" else n_total / world)
                      if summ is not None else None)
    ranks_seen = comm_info[0] if comm_info is not None else (int(round(ranks_by_count)) if ranks_by_count is not None else world)

    out = None
    if rank == 0:
        value = float(n_total) * args.steps / elapsed
        out = {
            "metric": "MCMC transitions/sec (whole node), 100-dim Gaussian, 65k chains",
            "value": value, "unit": "transitions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: MALA driftstep=0.9, lt=-|x|^2, D=100, 65,536 chains per GPU, "
                                   "VanillaMCTuner, x0~N(0,I); every step saved into per-chain running sums (save rule of run(job))"
                                   if not args.no_save else
                                   "BASELINE configs[1] without the save rule: MALA driftstep=0.9, lt=-|x|^2, D=100, VanillaMCTuner, x0~N(0,I)",
                       "nchains_per_gpu": n, "nchains_total": n_total, "ndims": NDIMS, "steps_per_launch": spl,
                       "save_rule": "running sums (KLARA_MON_SUMMARIES), burnin 0, thinning 1" if monitor else "off",
                       "running_sums_mode": "decided by the library on the device, launch by launch (no caller hint)",
                       "launches_4lane_8lane_device_decided": [int(v) for v in launch_counts],
                       "parallelism": f"chains sharded over {world} GPU(s), no data-path collective; summaries pooled on device"
                                      + (" and all-reduced over RCCL through the C ABI (klara_comm_init / klara_gather_moments)" if comm is not None else
                                         (" and all-reduced by torch.distributed (Python mirror)" if world > 1 else "")),
                       "collective": {"requested": args.collective,
                                      "used": ("klara_gather_moments over the library's RCCL communicator (C ABI)" if comm is not None else
                                               (f"torch.distributed all_reduce ({args.torch_backend if tgroup is not None else args.backend})" if world > 1
                                                else "none (one rank: pooled on the device)")),
                                      "rendezvous": f"torch.distributed {args.backend} (id broadcast, barriers, times)" if world > 1 else "none",
                                      "comm_nranks_rank_device": list(comm_info) if comm_info is not None else None,
                                      "ranks_by_transition_count": ranks_by_count, "error": collective_error, "torch_group_error": tgroup_error,
                                      "ab": gather_ab},
                       "streams": "library default (2 chain partitions on 2 HIP streams)" if args.streams == 0 else args.streams,
                       "timed_region": f"median of {args.reps} repetitions of {args.steps} transitions",
                       "repetition_ms_per_step": [t_ * 1e3 / args.steps for t_ in times],
                       "per_rank_ms_per_step": [statistics.median(rt[r] for rt in rank_times) * 1e3 / args.steps for r in range(world)],
                       "rank_time_max_over_min": max(statistics.median(rt[r] for rt in rank_times) for r in range(world))
                                                 / min(statistics.median(rt[r] for rt in rank_times) for r in range(world)),
                       "acceptance_rate": acc_rate, "rccl_ranks_seen": ranks_seen, "summary_gather_ms": gather_ms,
                       "device_clock_warmup": f"{args.clock_warmup} transitions of an identical scratch job before the timed repetitions" if args.clock_warmup > 0 else "off",
                       "timed_region_kernel_ms_per_step": statistics.median(kernel_ms_per_step),
                       "repetition_kernel_ms_per_step": kernel_ms_per_step},
        }
    eng.close()
    if comm is not None:
        comm.close()

    if rank == 0:
        out["roofline"] = rf = roofline_pass(K, L, n, spl, monitor, offset, local_rank, stream)
        # the same fraction over the TIMED REGION itself (HIP events around each repetition's K steps on this rank): there the library runs
        # the job as two chain partitions on two streams whose kernels overlap, so no single launch has a duration of its own — work / elapsed
        if rf.get("necessary_valu_insts_per_launch") and rf.get("transitions_per_launch"):
            per_step = rf["necessary_valu_insts_per_launch"] / rf["transitions_per_launch"]
            rf["frac_timed_region"] = 4.0 * per_step / (statistics.median(kernel_ms_per_step) * 1e-3) / rf["peak"]
            rf["frac_timed_region_source"] = ("necessary vector instructions per transition of this rank's chains x 4 / (klara_last_run_ms over the timed "
                                              "repetitions / steps) / peak: multi-launch runs overlap two chain partitions on two streams (a launch's ramp, its last "
                                              "round of wavefronts and the gap to the next launch are filled by the other partition), `frac` prices one launch alone")
    if rank == 0 and world == 1 and not args.no_extra:
        out["extra"] = extra_measurements(K, L, n, stream)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(L)
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()             # rank 0's roofline pass is over: every rank leaves the job together
        dist.destroy_process_group()
    if comm_abandoned:             # a helper thread is still inside RCCL's bootstrap: leave without waiting for it
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(0)


def diagt_kernel_name(sampler_id, lay_g, lay_e, onestep, unitw, mon, tune=False, da=False):
    b = lambda v: "true" if v else "false"
    return f"k_diagt<{sampler_id}, {lay_e // 2}, {lay_g}, {b(onestep)}, {b(unitw)}, {b(mon)}, {b(tune)}, {b(da)},"      # (prefix: further template flags follow)


def diagt_grid(n, lay_g):
    """threads of one whole-job launch of the pair-transposed kernels: one wavefront per group of 64 / lay_g chains, 4 per block"""
    cpw = 64 // lay_g
    return 256 * ((((n + cpw - 1) // cpw) + 3) // 4)


def launch_duration(e, spl, nlaunch=48, warm=8):
    """Mean duration of one launch: HIP events (klara_last_run_ms) around `nlaunch` back-to-back launches on ONE stream."""
    e.run(warm * spl)
    e.run(nlaunch * spl)
    kernel_ms, nl = e.last_run_ms()
    return kernel_ms * 1e-3 / max(nl, 1), nl


def roofline_pass(K, L, n, spl, monitor, offset, local_rank, stream):
    """The dominant kernel, one launch at a time: same workload with every launch on the caller's stream (nstreams = 1; the timed
    region overlaps two half-size launches on two streams, which says nothing about a single launch)."""
    e = K.Engine(sampler=L.SAMPLER_MALA, target=K.GaussDiagTarget.negdot(NDIMS), nchains=n, nsteps=10 ** 7, burnin=0,
                 driftstep=0.9, seed=20260927, chain_offset=offset, device=local_rank, monitor=monitor, steps_per_launch=spl,
                 stream=stream, nstreams=1)
    e.init_state_normal()
    launch_s, nlaunch = launch_duration(e, spl, nlaunch=64 if spl > 1 else 256)
    lay_kind, lay_g, lay_e = e.layout()
    _, _, nacc, ntr, _ = e.pooled_summaries(with_sums=False)
    cnt, last_mode, _ = e.launch_modes() if hasattr(L.load(), 'klara_get_launch_modes') else (None, [0], None)
    four = lay_kind == 3 and (not monitor or last_mode[0] == 0)     # the 4-lane kernels ran (they sum in the layout's 8-lane order)
    attrs = e.kernel_attributes(0 if four else 1, spl) if hasattr(L.load(), 'klara_get_kernel_attributes') else None
    clock_mhz = e.shader_clock_mhz() if hasattr(e, "shader_clock_mhz") else 0.0      # during the last of those launches (in-kernel probe)
    e.close()
    if four:
        lay_g, lay_e = 4, 2 * ((NDIMS + 7) // 8)
    if lay_kind == 3:
        kname = diagt_kernel_name(1, lay_g, lay_e, spl == 1 and not monitor, True, bool(monitor))
        label = (f"{kname} (pair-transposed layout: {lay_g} lanes x {lay_e // 2} element pairs per chain, {64 // lay_g} chains per "
                 f"wavefront; {spl} transitions per launch" + (", running sums" if monitor else "") + ")")
    else:
        kname, label = "k_transitions<1, 0,", f"k_transitions<MALA, GAUSS_DIAG, E={lay_e}>"
    grid = diagt_grid(n, lay_g) if lay_kind == 3 else None
    bud = budgets()["headline_4lane" if lay_g == 4 else "headline_8lane"] if lay_kind == 3 else None
    nwaves = (n + (64 // lay_g) - 1) // (64 // lay_g) if lay_kind == 3 else None
    rf = valu_roofline(kname, launch_s, grid=grid, label=label, attrs=attrs, budget=bud,
                       necessary_per_launch=bud["per_wave_transition"] * nwaves * spl if bud else None)
    rf.update(launches=nlaunch, chains_per_launch=n, transitions_per_launch=spl)
    if clock_mhz > 0.0 and rf.get("frac") is not None:
        # the peak above is priced at the 2.4 GHz the data sheet names; under this instruction mix the chip clocks lower (power), and
        # issue cycles that did not exist cannot be used: the same fraction against the cycles the SIMDs actually had
        rf["shader_clock_mhz"] = clock_mhz
        rf["frac_at_measured_clock"] = rf["frac"] * (CLOCK_HZ / (clock_mhz * 1e6))
        if rf.get("frac_at_measured_instruction_costs") is not None:
            rf["frac_at_measured_clock_and_instruction_costs"] = rf["frac_at_measured_instruction_costs"] * (CLOCK_HZ / (clock_mhz * 1e6))
    # the byte side.  S = 2*D*8 + 8 (x, gradient, log-target); SURVEY 8(d): B_K = (2 S + 1) / K per transition and chain.
    s_state = 2 * NDIMS * 8 + 8
    contract = n * (2 * s_state + 1)                               # per launch of K fused transitions: state in, state out, accepts
    # what this kernel has to move per launch: x, lt (and the held count) in — the gradient is re-formed from x; for the chains
    # that moved during the launch: x, gradient, lt and the accept counter out and, with the save rule on, their running sums read
    # and written once (sojourn form: a chain that did not move touches neither array)
    moved = min(1.0, 1.0 - (1.0 - nacc / max(ntr, 1)) ** spl)
    minimal = n * ((NDIMS * 8 + 8 + (16 if monitor else 0)) + moved * (s_state + 16 + (4 * NDIMS * 8 if monitor else 0)))
    hbm = {"contract_2S_plus_1_bytes_per_launch": contract, "minimal_bytes_per_launch": minimal,
           "contract_frac_of_8TBs": contract / launch_s / 1e9 / HBM_PEAK_GBS, "minimal_frac_of_8TBs": minimal / launch_s / 1e9 / HBM_PEAK_GBS,
           "fraction_of_chains_that_moved": moved, "traffic_bytes_per_launch": None}
    row = pmc_lookup(kname, grid)
    if row is not None and "FETCH_SIZE" in row["counters"] and "WRITE_SIZE" in row["counters"]:
        tb = (2.0 * row["counters"]["FETCH_SIZE"]["mean"] + row["counters"]["WRITE_SIZE"]["mean"]) * 1024.0
        hbm.update(traffic_bytes_per_launch=tb, traffic_frac_of_8TBs=tb / launch_s / 1e9 / HBM_PEAK_GBS, traffic_over_minimal=tb / minimal,
                   traffic_source=f"{PMC_JSON.relative_to(ROOT)}: FETCH_SIZE x 2 + WRITE_SIZE (KiB), separate --pmc passes; calibrated on "
                                  "the init kernel (profiles/README.md)")
    rf["traffic"] = hbm["traffic_bytes_per_launch"]
    rf["hbm"] = hbm
    if bud:
        rf["note"] = (f"VALU-issue bound: {bud['per_wave_transition']:.0f} necessary vector instructions per wavefront ({64 // lay_g} chains) and transition = "
                      f"{bud['per_wave_transition'] / (64 // lay_g):.1f} per chain and transition, {100.0 * bud['pair_evaluations_per_lane'] * 102 / bud['per_wave_transition']:.0f} % of them "
                      "the in-kernel Philox4x32-10 + Box-Muller of the proposal normals; HBM moves a few per cent of its peak (hbm.*)")
    return rf


def timed_rate(e, n, warm, steps):
    e.run(warm)
    t0 = time.perf_counter(); e.run(steps); dt = time.perf_counter() - t0
    ms, nl = e.last_run_ms()
    return n * steps / dt, ms * 1e-3 / max(nl, 1), nl


def extra_measurements(K, L, n, stream):
    """Outside the timed region: the other launch modes of the headline workload and the other BASELINE configurations, each with
    the roofline that bounds its kernel."""
    import numpy as np
    ex = {}
    bud = budgets()
    attrs_of = lambda e, nsteps, which=0: e.kernel_attributes(which, nsteps) if hasattr(L.load(), "klara_get_kernel_attributes") else None
    neg = K.GaussDiagTarget.negdot(NDIMS)
    # -- the headline workload in its other modes
    for key, kw in (("mala_one_transition_per_launch_no_save", dict(steps_per_launch=1, monitor=0)),
                    ("mala_fused_no_save", dict(steps_per_launch=0, monitor=0)),
                    ("mala_one_transition_per_launch_with_save", dict(steps_per_launch=1, monitor=L.MON_SUMMARIES)),
                    ("mala_fused_with_save_4lane_kernels_forced", dict(steps_per_launch=0, monitor=L.MON_SUMMARIES, sparse_moves=1)),
                    ("mala_fused_with_save_8lane_kernels_forced", dict(steps_per_launch=0, monitor=L.MON_SUMMARIES, sparse_moves=2))):
        e = K.Engine(sampler=L.SAMPLER_MALA, target=neg, nchains=n, nsteps=10 ** 7, driftstep=0.9, stream=stream, **kw)
        e.init_state_normal()
        rate, _, _ = timed_rate(e, n, 64, 1024)
        ex[f"{key}_transitions_per_s"] = rate
        e.close()
    # -- the same sampler at a step that MIXES (at driftstep 0.9 the headline job rejects 99 % of its proposals, so its timed region
    # hardly runs the commit / fold path): driftstep 0.3 -> acceptance 0.56 (AcceptanceRateMCTuner's target for MALA is 0.574), running sums on, the
    # kernel family decided by the library (resident sums on 8 lanes per chain at this acceptance)
    e = K.Engine(sampler=L.SAMPLER_MALA, target=neg, nchains=n, nsteps=10 ** 7, driftstep=0.3, stream=stream, monitor=L.MON_SUMMARIES)
    e.init_state_normal()
    rate, _, _ = timed_rate(e, n, 256, 1024)
    _, _, nacc, ntr, _ = e.pooled_summaries(with_sums=False)
    ex["mala_d100_mixing_step_transitions_per_s"] = rate
    ex["mala_d100_mixing_step"] = {"driftstep": 0.3, "acceptance_rate": nacc / max(ntr, 1), "save_rule": "running sums",
                                   "launches_4lane_8lane_device_decided": [int(v) for v in e.launch_modes()[0]]}
    e.close()
    e = K.Engine(sampler=L.SAMPLER_MALA, target=neg, nchains=n, nsteps=10 ** 7, driftstep=0.3, stream=stream, monitor=L.MON_SUMMARIES, nstreams=1)
    e.init_state_normal()
    ls, _ = launch_duration(e, L.DEFAULT_STEPS_PER_LAUNCH, nlaunch=32)
    lay_m = e.layout(); last_mode = e.launch_modes()[1]
    four_m = last_mode[0] == 0
    at = attrs_of(e, L.DEFAULT_STEPS_PER_LAUNCH, 0 if four_m else 1); e.close()
    bm = bud["headline_4lane" if four_m else "headline_8lane"]
    gm, em = (4, 2 * ((NDIMS + 7) // 8)) if four_m else (lay_m[1], lay_m[2])
    ex["mala_d100_mixing_step_roofline"] = valu_roofline(diagt_kernel_name(1, gm, em, False, True, True), ls, grid=diagt_grid(n, gm), attrs=at, budget=bm,
                                                         necessary_per_launch=bm["per_wave_transition"] * ((n + bm["chains_per_wave"] - 1) // bm["chains_per_wave"]) * L.DEFAULT_STEPS_PER_LAUNCH)
    ex["mala_d100_mixing_step_roofline"]["note"] = ("the budget prices the transition (normals, proposal, ratio, accept test), not the commit of an accepted proposal nor the fold of "
                                                    "the running sums, which this job executes on every second transition of every chain")

    # one launch at a time for the single-transition kernel (round 1's roofline kernel)
    e = K.Engine(sampler=L.SAMPLER_MALA, target=neg, nchains=n, nsteps=10 ** 7, driftstep=0.9, stream=stream, steps_per_launch=1,
                 monitor=0, nstreams=1)
    e.init_state_normal()
    ls, _ = launch_duration(e, 1, nlaunch=256)
    at = attrs_of(e, 1); e.close()
    b4 = bud["headline_4lane"]                          # (the 4-lane kernels run every job that keeps no running sums)
    ex["mala_one_transition_per_launch_roofline"] = valu_roofline(diagt_kernel_name(1, 4, 2 * ((NDIMS + 7) // 8), True, True, False), ls, grid=diagt_grid(n, 4),
                                                                  attrs=at, necessary_per_launch=(b4["per_wave_transition"] - b4["bookkeeping"]) * ((n + 15) // 16))

    # -- cfg 1: the README job (README.md:23-47: MH, sigma = (1, 1), lt = -dot(z, z), D = 2, burn-in 1000, x0 = (5.1, -0.9)) as 1,048,576
    # replicas — one chain per lane — with the running sums of mean(chain) on
    nr = 1 << 20
    e = K.Engine(sampler=L.SAMPLER_MH, target=K.GaussDiagTarget.negdot(2), nchains=nr, nsteps=10 ** 7, burnin=1000, mh_sigma=[1.0, 1.0],
                 monitor=L.MON_SUMMARIES, stream=stream, nstreams=1)
    e.set_state(np.tile([5.1, -0.9], (nr, 1)))
    rate, ls, _ = timed_rate(e, nr, 1024, 2048)
    at = attrs_of(e, 32); e.close()
    ex["cfg1_readme_mh_1048576_replicas_transitions_per_s"] = rate
    b1 = bud["cfg1"]           # (launch = the library's 32 transitions of the 1,048,576 chains: 16,384 wavefronts of 64 chains)
    ex["cfg1_roofline"] = valu_roofline("k_transitions<0, 0, 2, 0, 1>", ls, attrs=at, budget=b1,
                                        necessary_per_launch=b1["per_wave_transition"] * (nr // b1["chains_per_wave"]) * L.DEFAULT_STEPS_PER_LAUNCH)

    # -- HMC L=10 eps=0.1 on the README target (VALU) and on the dense target (FP64 MFMA; cfg 3)
    e = K.Engine(sampler=L.SAMPLER_HMC, target=neg, nchains=n, nsteps=10 ** 7, leapstep=0.1, nleaps=10, stream=stream, nstreams=1)
    e.init_state_normal()
    rate, ls, _ = timed_rate(e, n, 512, 512)      # (~13 ms of warm-up: a device that idled while the job was created ramps its clocks for ~20 ms)
    lay = e.layout(); at = attrs_of(e, 32); e.close()
    ex["hmc_iso_leapfrog_chain_per_s"] = rate * 10
    # (round 5: an unmonitored HMC job on this target runs the 4-lanes-per-chain kernels, NP = ceil(D / 8) pairs per lane, which sum in the layout's 8-lane order)
    four = NDIMS <= 104 and "hmc_iso_4lane" in bud
    bh = bud["hmc_iso_4lane"] if four else bud["hmc_iso"]
    hg, he = (4, 2 * ((NDIMS + 7) // 8)) if four else (lay[1], lay[2])
    ex["hmc_iso_roofline"] = valu_roofline(diagt_kernel_name(2, hg, he, False, True, False), ls, grid=diagt_grid(n, hg), attrs=at, budget=bh,
                                           necessary_per_launch=bh["per_wave_transition"] * (n // bh["chains_per_wave"]) * L.DEFAULT_STEPS_PER_LAUNCH)

    e = K.Engine(sampler=L.SAMPLER_HMC, target=K.GaussDenseTarget.compound_symmetric(NDIMS, 0.5), nchains=n, nsteps=10 ** 7,
                 leapstep=0.1, nleaps=10, stream=stream)
    e.init_state_normal()
    rate, ls, _ = timed_rate(e, n, 32, 128)
    at = attrs_of(e, L.DEFAULT_STEPS_PER_LAUNCH); e.close()
    flops_per_launch = n * L.DEFAULT_STEPS_PER_LAUNCH * 10 * (2 * NDIMS * NDIMS + 6 * NDIMS)         # 2 D^2 + 6 D per leapfrog and chain
    tf = flops_per_launch / ls / 1e12
    ex["cfg3_hmc_dense_leapfrog_chain_per_s"] = rate * 10
    ex["cfg3_hmc_dense_transitions_per_s"] = rate
    rf = {"bound": "mfma", "achieved": tf, "peak": FP64_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": tf / FP64_MFMA_PEAK_TF,
          "kernel": f"k_dense_transitions<HMC, NE=25> (v_mfma_f64_16x16x4 + v_mfma_f64_4x4x4_4b tail tile), {L.DEFAULT_STEPS_PER_LAUNCH} transitions per launch",
          "launch_us": ls * 1e6,
          "source": "algorithmic flops (2 D^2 + 6 D per leapfrog and chain, SURVEY 8(d)) / launch duration from HIP events in this run"}
    row = pmc_lookup("k_dense_transitions<2")
    rec = pmc_recorded_attributes().get("k_dense_transitions<2")
    rf["pmc"] = {"file": f"profiles/{PMC_JSON.name}", "key": "k_dense_transitions<2", "kernel": row["kernel"] if row else None,
                 "loaded_vgpr": 8 * ((at[0] + 7) // 8) if at else None, "loaded_scratch": at[1] if at else None,
                 "stale": bool(at is not None and rec is not None and [8 * ((at[0] + 7) // 8), at[1]] != list(rec)) if (at and rec) else None}
    if row is not None and "SQ_VALU_MFMA_BUSY_CYCLES" in row["counters"] and not rf["pmc"]["stale"]:
        busy = row["counters"]["SQ_VALU_MFMA_BUSY_CYCLES"]["mean"]
        rf["mfma_pipe_busy_frac"] = busy / NSIMD / (ls * CLOCK_HZ)
        rf["mfma_pipe_source"] = f"SQ_VALU_MFMA_BUSY_CYCLES per launch ({PMC_JSON.relative_to(ROOT)}) / 1024 SIMDs / (launch duration x 2.4 GHz)"
    ex["cfg3_hmc_dense_roofline"] = rf

    # -- the same sampler on a dense 256 x 256 precision: beyond what fits the LDS, P streamed from L2 through a register ring, the momentum in
    # LDS, one wavefront per SIMD (klara_dense_big.h; round 4)
    try:
        D2 = 256
        e = K.Engine(sampler=L.SAMPLER_HMC, target=K.GaussDenseTarget.compound_symmetric(D2, 0.5), nchains=n, nsteps=10 ** 7,
                     leapstep=0.05, nleaps=10, steps_per_launch=16, stream=stream)
        e.init_state_normal()
        rate, ls, _ = timed_rate(e, n, 16, 32)
        lay2 = e.layout(); e.close()
        tf2 = n * 16 * 10 * (2 * D2 * D2 + 6 * D2) / ls / 1e12
        ex["hmc_dense_d256_leapfrog_chain_per_s"] = rate * 10
        ex["hmc_dense_d256_roofline"] = {"bound": "mfma", "achieved": tf2, "peak": FP64_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": tf2 / FP64_MFMA_PEAK_TF,
                                         "kernel": f"k_dense_hmc_big<NE={lay2[2]}> (v_mfma_f64_16x16x4, A fragments streamed from memory), 16 transitions per launch",
                                         "launch_us": ls * 1e6,
                                         "source": "algorithmic flops (2 D^2 + 6 D per leapfrog and chain) / launch duration from HIP events in this run"}
    except Exception as exc:
        ex["hmc_dense_d256_error"] = repr(exc)
    # ... and MALA / MH on the same target (one gradient per transition: 2 D^2 flop per transition and chain; round 5: the current gradient stays in the
    # accumulators, the current value and MH's proposal scales in LDS)
    try:
        D2 = 256
        for key, kw in (("mala", dict(sampler=L.SAMPLER_MALA, driftstep=0.002)), ("mh", dict(sampler=L.SAMPLER_MH, mh_sigma=np.full(D2, 0.02)))):
            e = K.Engine(target=K.GaussDenseTarget.compound_symmetric(D2, 0.5), nchains=n, nsteps=10 ** 7, steps_per_launch=32, stream=stream, **kw)
            e.init_state_normal()
            rate, ls, _ = timed_rate(e, n, 64, 128)
            lay2 = e.layout(); e.close()
            tf2 = n * 32 * 2 * D2 * D2 / ls / 1e12
            ex[f"{key}_dense_d256_transitions_per_s"] = rate
            ex[f"{key}_dense_d256_roofline"] = {"bound": "mfma", "achieved": tf2, "peak": FP64_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": tf2 / FP64_MFMA_PEAK_TF,
                                                "kernel": f"k_dense_big<{key.upper()}, NE={lay2[2]}> (v_mfma_f64_16x16x4, A fragments streamed from memory), 32 transitions per launch",
                                                "launch_us": ls * 1e6,
                                                "source": "algorithmic flops (2 D^2 per transition and chain: one gradient) / launch duration from HIP events in this run"}
    except Exception as exc:
        ex["mala_mh_dense_d256_error"] = repr(exc)

    # -- dense targets beyond D = 256 (round 6; refused before): the tile of 16 chains on a workgroup of 8 / 16 wavefronts that deal the row tiles of P evenly,
    # the proposal exchanged through LDS (klara_dense_split.h, layout kind 6); 2 D^2 flop per gradient and chain
    try:
        for D2 in (512, 1024):
            for key, grads, spl, kw in (("hmc", 10, 4, dict(sampler=L.SAMPLER_HMC, leapstep=0.1 * (256 / D2) ** 0.25, nleaps=10)),
                                        ("mala", 1, 32, dict(sampler=L.SAMPLER_MALA, driftstep=0.002 * 256 / D2))):
                e = K.Engine(target=K.GaussDenseTarget.compound_symmetric(D2, 0.5), nchains=n, nsteps=10 ** 7, steps_per_launch=spl, stream=stream, **kw)
                e.init_state_normal()
                rate, ls, _ = timed_rate(e, n, spl, 2 * spl)
                lay2 = e.layout(); e.close()
                tf2 = n * spl * grads * 2 * D2 * D2 / ls / 1e12
                ex[f"{key}_dense_d{D2}_" + ("leapfrog_chain_per_s" if grads > 1 else "transitions_per_s")] = rate * grads
                ex[f"{key}_dense_d{D2}_roofline"] = {"bound": "mfma", "achieved": tf2, "peak": FP64_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": tf2 / FP64_MFMA_PEAK_TF,
                                                    "kernel": f"k_dense_split<{key.upper()}> (layout kind {lay2[0]}: {lay2[1]} wavefronts per tile of 16 chains; v_mfma_f64_16x16x4, A fragments "
                                                              f"streamed from memory, x exchanged through LDS), {spl} transitions per launch",
                                                    "launch_us": ls * 1e6,
                                                    "source": "algorithmic flops (2 D^2 per gradient and chain) / launch duration from HIP events in this run"}
    except Exception as exc:
        ex["dense_split_error"] = repr(exc)

    # -- slice sampler on the README target, D = 100: the library's own launch length for this job (KLARA_DEFAULT_STEPS_PER_LAUNCH_SLICE; the lanes run out of
    # lockstep and a wavefront waits for its slowest lane once per element slot and launch, klara_diagt_slice.h)
    SLICE_SPL = L.DEFAULT_STEPS_PER_LAUNCH_SLICE
    e = K.Engine(sampler=L.SAMPLER_SLICE, target=neg, nchains=n, nsteps=10 ** 7, slice_widths=np.full(NDIMS, 1.0), steps_per_launch=0,
                 stream=stream, nstreams=1)
    e.init_state_normal()
    rate, ls, _ = timed_rate(e, n, SLICE_SPL, 4 * SLICE_SPL)
    lay = e.layout(); at = attrs_of(e, SLICE_SPL); e.close()
    ex["slice_d100_transitions_per_s"] = rate
    ex["slice_d100_coordinate_updates_per_s"] = rate * NDIMS
    # the slice sampler's trip counts are data dependent: the algorithmic budget takes an update's own mean probe counts on this target in
    # stationarity (a seeded simulation of the procedure, scripts/instruction_budget.py slice_probe_counts); `frac_lockstep` prices what 64 coordinate
    # updates that share their loops have to execute — the mean of the maximum over 64 (round 4's kernel, still the one for jobs that keep a history)
    bs, bsl = bud["slice_d100"], bud["slice_d100_lockstep"]
    ex["slice_d100_roofline"] = valu_roofline(f"k_diagt_slice_free<{lay[1]},", ls, attrs=at, budget=bs,
                                              necessary_per_launch=bs["per_wave_transition"] * (n // bs["chains_per_wave"]) * SLICE_SPL)
    ex["slice_d100_roofline"]["transitions_per_launch"] = SLICE_SPL
    ex["slice_d100_roofline"]["frac_lockstep"] = 4.0 * bsl["per_wave_transition"] * (n // bsl["chains_per_wave"]) * SLICE_SPL / ls / (NSIMD * CLOCK_HZ)

    # ... and on the dense 100 x 100 precision of cfg 3: a probe is a full evaluation = one matrix pass over the 16 chains of a tile, each chain taking from
    # it the probe its own coordinate and stage ask for (round 5: slice_dense_free, klara_dense.h); beyond D = 128 on the streamed layouts
    try:
        for key, dd, nn in (("slice_dense_d100", NDIMS, n), ("slice_dense_d256", 256, 16384)):
            e = K.Engine(sampler=L.SAMPLER_SLICE, target=K.GaussDenseTarget.compound_symmetric(dd, 0.5), nchains=nn, nsteps=10 ** 7, slice_widths=np.full(dd, 2.0),
                         steps_per_launch=1, stream=stream)
            e.init_state_normal()
            rate, ls, _ = timed_rate(e, nn, 1, 2)
            lay = e.layout(); e.close()
            ex[f"{key}_chain_transitions_per_s"] = rate
            ex[f"{key}_coordinate_updates_per_s"] = rate * dd
            ex[f"{key}_layout"] = list(lay)
    except Exception as exc:
        ex["slice_dense_error"] = repr(exc)

    # -- the slice sampler on GENERAL targets: every probe is a full evaluation of the log-target by the chain's lanes
    # (SliceSampler.jl:77-94 as written; the difference form above exists for diagonal Gaussians only).  The swiss logistic regression
    # (doc/examples/swiss/SliceSampler.jl: D = 4, a probe = a pass over the 200 data rows) and a pair closure (the README target written as
    # klara_user_pair, D = 100: run as a whole-vector closure staged through LDS)
    try:
        gold = ROOT / "tests" / "golden"
        sw = np.load(gold / "swiss.npz")
        Xs = sw["measurements"]; Xs = np.ascontiguousarray((Xs - Xs.mean(axis=0)) / Xs.std(axis=0, ddof=1))
        ys = np.ascontiguousarray(sw["status"].astype(np.float64))
        nc = 32768
        e = K.Engine(sampler=L.SAMPLER_SLICE, target=K.LogisticTarget(Xs, ys, 100.0), nchains=nc, nsteps=10 ** 6, slice_widths=np.full(4, 1.0), stream=stream)
        e.set_state(0.1 * np.random.default_rng(3).standard_normal((nc, 4)))
        rate, ls, _ = timed_rate(e, nc, 64, 128)
        ex["slice_swiss_logistic_chain_transitions_per_s"] = rate
        ex["slice_swiss_logistic_coordinate_updates_per_s"] = rate * 4
        ex["slice_swiss_logistic_layout"] = list(e.layout()); e.close()
        src_pair = ("KLARA_USER_FN double klara_user_pair(double x0, double x1, int pair, int D, const double* data, long long ndata, double* g0, double* g1)\n"
                    "{ *g0 = -2.0 * x0; *g1 = -2.0 * x1; return -(x0 * x0) - (x1 * x1); }\n")
        # (round 6: on the few-lanes kernels, a probe compares the pair's own term — k_diagt<SLICE, .., USERPAIR>; round 5 summed the pairs and ran a whole-vector closure: 7.4e8)
        e = K.Engine(sampler=L.SAMPLER_SLICE, target=K.CustomTarget.pairwise(NDIMS, src_pair), nchains=n, nsteps=10 ** 6, slice_widths=np.full(NDIMS, 1.0),
                     steps_per_launch=0, stream=stream)
        e.init_state_normal()
        rate, ls, _ = timed_rate(e, n, 32, 64)
        ex["slice_pair_closure_d100_chain_transitions_per_s"] = rate
        ex["slice_pair_closure_d100_coordinate_updates_per_s"] = rate * NDIMS
        ex["slice_pair_closure_d100_layout"] = list(e.layout()); e.close()
    except Exception as exc:
        ex["slice_general_target_error"] = repr(exc)

    # -- the logistic regression BEYOND 16 parameters (round 6, klara_logit_mfma.h): X p and X'(y - logistic(Xp)) of 16 chains per wavefront on the FP64
    # matrix cores, X streamed from memory (rounds 1-5: a closure, one chain per lane).  64 parameters, 200 synthetic rows, 32,768 chains, running sums;
    # algorithmic flops: 4 n D per gradient evaluation and chain (two passes of 2 n D)
    try:
        dl, nl, ncl = 64, 200, 32768
        rngl = np.random.default_rng(64)
        Xl = rngl.standard_normal((nl, dl)); bl = rngl.standard_normal(dl)
        yl = (rngl.random(nl) < 1.0 / (1.0 + np.exp(-Xl @ bl / np.sqrt(dl)))).astype(np.float64)
        tl_ = K.LogisticTarget(Xl / np.sqrt(dl), yl, 10.0)
        x0l = 0.1 * rngl.standard_normal((ncl, dl))
        for key, kw, evals in (("mala", dict(sampler=L.SAMPLER_MALA, driftstep=0.05), 1), ("hmc_L10", dict(sampler=L.SAMPLER_HMC, leapstep=0.05, nleaps=10), 10)):
            e = K.Engine(target=tl_, nchains=ncl, nsteps=10 ** 6, monitor=L.MON_SUMMARIES, steps_per_launch=8, stream=stream, **kw)
            e.set_state(x0l)
            rate, ls, _ = timed_rate(e, ncl, 16, 64)
            layl = e.layout(); e.close()
            tfl = ncl * 8 * evals * 4.0 * nl * dl / ls / 1e12
            ex[f"logistic_d64_n200_{key}_transitions_per_s"] = rate
            ex[f"logistic_d64_n200_{key}_roofline"] = {"bound": "mfma", "achieved": tfl, "peak": FP64_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": tfl / FP64_MFMA_PEAK_TF,
                                                       "kernel": f"k_logit_mfma<{key.split('_')[0].upper()}, NE={layl[2]}> (v_mfma_f64_16x16x4, fragments of X streamed), 8 transitions per launch",
                                                       "launch_us": ls * 1e6, "layout": list(layl),
                                                       "source": "algorithmic flops (4 n D per gradient evaluation and chain) / launch duration from HIP events in this run; the rows' "
                                                                 "transcendental arithmetic (~60 vector instructions per row and chain) shares the SIMD with the matrix passes"}
    except Exception as exc:
        ex["logistic_mfma_error"] = repr(exc)

    # -- the two data-model configurations of BASELINE.json at their per-GPU share (cfg 4: 262,144 / 8 chains of the swiss
    # logistic regression, MALA h = 0.1; cfg 5: 1,048,576 / 8 chains of the rats hierarchical model, HMC L = 32 with the
    # per-GPU pooled AcceptanceRate tuner).  Data: the reference's own files as committed fixtures (tests/golden/*.npz).
    try:
        gold = ROOT / "tests" / "golden"
        sw = np.load(gold / "swiss.npz")
        X = sw["measurements"]; X = np.ascontiguousarray((X - X.mean(axis=0)) / X.std(axis=0, ddof=1))
        y = np.ascontiguousarray(sw["status"].astype(np.float64))
        nc = 32768
        x0 = np.array([5.1, -0.9, 8.2, -4.5])[None, :] + 0.1 * np.random.default_rng(0).standard_normal((nc, 4))
        e = K.Engine(sampler=L.SAMPLER_MALA, target=K.LogisticTarget(X, y, 100.0), nchains=nc, nsteps=10 ** 6, burnin=1000, driftstep=0.1,
                     steps_per_launch=50, monitor=L.MON_SUMMARIES, stream=stream)
        e.set_state(x0)
        rate, ls, _ = timed_rate(e, nc, 1000, 1000)      # (~16 ms of warm-up, see above: the first 500 transitions of this job run 10 % below its steady rate)
        at = attrs_of(e, 50); lay4 = e.layout(); e.close()
        ex["cfg4_swiss_logistic_mala_transitions_per_s_per_gpu"] = rate
        assert lay4[:2] == (2, 64 // bud["cfg4"]["chains_per_wave"]), lay4      # (the budget's lanes per chain are the job's row split)
        ex["cfg4_roofline"] = valu_roofline("k_transitions<1, 2,", ls, attrs=at, budget=bud["cfg4"],
                                            necessary_per_launch=bud["cfg4"]["per_wave_transition"] * (nc // bud["cfg4"]["chains_per_wave"]) * 50)
        rats = np.load(gold / "rats.npz")
        t = K.HierNormalTarget(rats["weight"], rats["age"] - 22.0)
        nc = 131072
        x0 = t.least_squares_start()[None, :] + 0.05 * np.random.default_rng(1).standard_normal((nc, t.ndims))
        e = K.Engine(sampler=L.SAMPLER_HMC, target=t, nchains=nc, nsteps=2000, burnin=1000, leapstep=0.02, nleaps=32,
                     tuner=L.TUNER_ACCEPT_RATE, tuner_mode=L.TUNE_POOLED, targetrate=0.65, period=100, steps_per_launch=10,
                     monitor=L.MON_SUMMARIES, stream=stream)
        e.set_state(x0)
        rate, ls, _ = timed_rate(e, nc, 100, 200)
        at = attrs_of(e, 10); e.close()
        ex["cfg5_rats_hmc_L32_leapfrog_chain_per_s_per_gpu"] = rate * 32
        ex["cfg5_rats_hmc_L32_transitions_per_s_per_gpu"] = rate
        ex["cfg5_roofline"] = valu_roofline("k_hiert<2,", ls, attrs=at, budget=bud["cfg5"],
                                            necessary_per_launch=bud["cfg5"]["per_wave_transition"] * (nc // bud["cfg5"]["chains_per_wave"]) * 10)
    except Exception as exc:      # the fixtures are part of the repository; a failure here must not lose the headline line
        ex["model_configs_error"] = repr(exc)
    return ex


def cpu_baseline(L):
    """CPU oracle (restatement of the reference path, OpenMP over chains) on a bounded sample of the workload (no save rule: the
    oracle's transition loop alone — the baseline is not charged for it)."""
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
           "sample": f"oracle/libklara_oracle.so (C restatement, gcc -O3 -mavx2 -mfma -ffp-contract=off, OpenMP over chains, {cores} threads): "
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
