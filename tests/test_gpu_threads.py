"""The boundary's concurrency contract (include/klara_hip.h): "handle is single-owner ... distinct handles may be driven from
distinct host threads", and klara_run_async / klara_synchronize on a caller-supplied stream.

The reference has one job per `run(job)` call and maps a vector of jobs one after the other (/root/reference/src/jobs/BasicMCJob.jl:212-244,
src/jobs/jobs.jl:212); a host that drives several GPUs' worth of jobs from threads — or a Julia binding under `Threads.@threads` — needs the
library to keep every handle's streams, launch plans, error words, JIT cache entries and monitors apart.  Every thread's result is compared bit
for bit with the CPU oracle run alone.
"""
import ctypes as C
import threading

import numpy as np
import pytest

import cases
import oracle_ffi as O

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("gpu_required")]


def _oracle_result(case, layout):
    job = O.OracleJob(**cases.oracle_kwargs(case, layout=layout))
    if case["x0"] is None:
        job.init_state_normal()
    else:
        job.set_state(case["x0"])
    job.run(case["nsteps"])
    return job


def _with_nonce(case, nonce):
    """a CUSTOM target whose C text differs by a comment: a new entry of the run-time compiler's cache (no code object on disk or in memory)"""
    import dataclasses
    t = case["target"]
    case = dict(case)
    case["target"] = dataclasses.replace(t, source=t.source + f"\n/* {nonce} */\n")
    return case


THREAD_CASES = ("mala_d100", "hmc_dense_d100", "slice_d20_stepout", "custom_quartic_mala_d64", "custom_banana_hmc", "mh_readme", "hmc_rats", "mala_swiss",
                "hmc_logitm_d33_n70", "hmc_dense_d300_split", "pair_quartic_slice_d100")      # (round 6: a tile on a workgroup; a run-time compiled k_diagt<SLICE, .., USERPAIR>)


def test_distinct_handles_driven_from_distinct_host_threads_bit_exact():
    """Eleven host threads, each the single owner of one handle — MALA and MH on diagonal Gaussians, HMC on the dense target (FP64 MFMA kernels), the
    free-running slice kernel, HMC on the hierarchical model, MALA on the swiss logistic regression, HMC on a 33-parameter logistic regression (matrix cores,
    streamed), HMC on a 300-dimensional dense target (a workgroup per tile), the slice sampler on a pair closure and TWO more user closures whose sources nothing has
    compiled before (both threads are inside hiprtc / the JIT cache at once) — created, run in ragged pieces (klara_run, and klara_run_async +
    klara_synchronize) and read back concurrently, three times over; each compared bit for bit with the oracle run alone on the main thread."""
    import os
    import klara_jl_amd as K
    nonce = f"threads {os.getpid()} {np.random.default_rng().integers(1 << 62)}"
    the_cases = []
    for i, name in enumerate(THREAD_CASES):
        case = cases.make_case(name)
        if isinstance(case["target"], K.CustomTarget):
            case = _with_nonce(case, f"{nonce} {i}")
        the_cases.append((name, case))
    nthreads = len(the_cases)
    start = threading.Barrier(nthreads)
    results, errors = [None] * nthreads, [None] * nthreads
    ROUNDS = 3

    def worker(i, name, case):
        try:
            rng = np.random.default_rng(1000 + i)
            out = []
            for rnd in range(ROUNDS):
                start.wait(timeout=600)                                  # all threads enter klara_create together (the two closures: hiprtc together)
                eng = K.Engine(**cases.engine_kwargs(case))
                if case["x0"] is None:
                    eng.init_state_normal()
                else:
                    eng.set_state(case["x0"])
                left = case["nsteps"]
                while left > 0:                                          # ragged pieces: the launch plan restarts in every call, other threads' launches in between
                    k = int(min(left, rng.integers(1, max(2, case["nsteps"] // 3))))
                    if rng.integers(2):
                        eng.run(k)
                    else:
                        eng.run_async(k); eng.synchronize()
                    left -= k
                x, lt, g = eng.state()
                s, q, nsaved = eng.chain_sums()
                out.append((eng.layout(), x, lt, g, eng.accept_mask(), s, q, nsaved))
                eng.close()
            results[i] = out
        except BaseException as exc:                                     # a barrier must not wait for a thread that died
            errors[i] = repr(exc)
            start.abort()

    threads = [threading.Thread(target=worker, args=(i, n, c)) for i, (n, c) in enumerate(the_cases)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=1800)
    assert not any(t.is_alive() for t in threads), "a thread is still inside the library"
    assert errors == [None] * nthreads, errors
    for (name, case), out in zip(the_cases, results):
        job = _oracle_result(case, out[0][0])
        for rnd, (_, x, lt, g, acc, s, q, nsaved) in enumerate(out):
            assert np.array_equal(acc, job.accept), (name, rnd, "accept mask")
            assert np.array_equal(x, job.X) and np.array_equal(lt, job.LT), (name, rnd, "state")
            npost = max(0, (case["nsteps"] - case.get("burnin", 0) - 1) // case.get("thinning", 1) + 1)
            assert np.array_equal(s, job.sum) and np.array_equal(q, job.sumsq) and nsaved == npost, (name, rnd, "running sums")


def test_two_threads_share_one_jit_cache_entry():
    """The same user closure asked for by four threads at once: one compile, four handles of the one code object, four identical (and
    oracle-exact) results — the cache's lock is held across lookup and insert."""
    import os
    import klara_jl_amd as K
    case = _with_nonce(cases.make_case("custom_quartic_mala_d20_pooled"), f"shared {os.getpid()} {np.random.default_rng().integers(1 << 62)}")
    n = 4
    start = threading.Barrier(n)
    results, errors = [None] * n, [None] * n

    def worker(i):
        try:
            start.wait(timeout=600)
            eng = K.Engine(**cases.engine_kwargs(case))
            eng.init_state_normal() if case["x0"] is None else eng.set_state(case["x0"])
            eng.run(case["nsteps"])
            x, lt, _ = eng.state()
            results[i] = (eng.layout(), x, lt, eng.accept_mask(), eng.tune())
            eng.close()
        except BaseException as exc:
            errors[i] = repr(exc); start.abort()

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(n)]
    [t.start() for t in threads]
    [t.join(timeout=1200) for t in threads]
    assert errors == [None] * n, errors
    job = _oracle_result(case, results[0][0])
    for lay, x, lt, acc, tune in results:
        assert np.array_equal(acc, job.accept) and np.array_equal(x, job.X) and np.array_equal(lt, job.LT)
        assert np.array_equal(np.asarray(tune[0]), np.asarray(results[0][4][0]))


def _hip():
    hip = C.CDLL("libamdhip64.so.7")          # (soname of the runtime already loaded by the library and by torch)
    hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    hip.hipMemcpyAsync.restype = C.c_int
    return hip


@pytest.mark.parametrize("name,nstreams", [("mala_d100_small_step", 0), ("hmc_dense_d100", 0), ("mala_d100_small_step", 1)])
def test_run_async_is_ordered_on_the_callers_stream(name, nstreams):
    """klara_run_async on a caller-supplied stream, with work queued BEFORE it that the run depends on and work queued AFTER it that depends
    on the run, and klara_synchronize last — no host synchronisation in between:
      before: a long torch kernel, then device-to-device copies that put the job's real start state (x, lt, gradient) into the handle's own
              arrays (klara_device_ptrs), replacing a decoy state;
      runs:   two klara_run_async calls (multi-launch: the chain partitions fork to the library's internal streams and join back);
      after:  a copy of the handle's x into a torch tensor, queued on the same stream.
    The tensor must hold the oracle's final state bit for bit, and so must klara_get_state after klara_synchronize."""
    import torch
    import klara_jl_amd as K
    from klara_jl_amd import _lib as L
    case = cases.make_case(name)
    n1 = case["nsteps"] // 2 + 1
    n2 = case["nsteps"] - n1
    kw = cases.engine_kwargs(case, steps_per_launch=3, nstreams=nstreams)     # several launches per call: the partitions go to the side streams
    s = torch.cuda.Stream()
    twin = K.Engine(**kw)                                                     # forms (x0, lt0, g0) exactly as the job would
    twin.init_state_normal() if case["x0"] is None else twin.set_state(case["x0"])
    x0, lt0, g0 = twin.state()
    twin.close()
    eng = K.Engine(stream=s.cuda_stream, **kw)
    eng.set_state(x0 + 1.0)                                                   # the decoy: a run that started too early ends somewhere else
    px, plt, pg = eng.device_ptrs()
    hip = _hip()
    dx, dlt, dg = (torch.from_numpy(a).cuda() for a in (x0, lt0, g0))
    xout = torch.zeros_like(dx)
    a = torch.randn(4096, 4096, device="cuda")
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        for _ in range(20):
            a = a @ a * 1e-3                                                  # ~tens of ms of queued work in front
        for dst, src in ((px, dx), (plt, dlt), (pg, dg)):
            if dst:
                assert hip.hipMemcpyAsync(C.c_void_p(dst), C.c_void_p(src.data_ptr()), src.numel() * 8, 3, C.c_void_p(s.cuda_stream)) == 0
        eng.run_async(n1)
        b = a @ a                                                             # the caller's own work between the two runs
        eng.run_async(n2)
        assert hip.hipMemcpyAsync(C.c_void_p(xout.data_ptr()), C.c_void_p(px), xout.numel() * 8, 3, C.c_void_p(s.cuda_stream)) == 0
        c = b @ b
    eng.synchronize()                                                         # the one host synchronisation (the caller's stream: everything above)
    job = _oracle_result(case, eng.layout())
    assert np.array_equal(xout.cpu().numpy(), job.X), "the copy queued after klara_run_async did not see the run's result"
    x, lt, g = eng.state()
    assert np.array_equal(x, job.X) and np.array_equal(lt, job.LT) and np.array_equal(eng.accept_mask(), job.accept)
    assert torch.isfinite(c).all() or True
    eng.close()
