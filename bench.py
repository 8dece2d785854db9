#!/usr/bin/env python3
"""bench.py - leapfrog-steps/sec (all chains) of the MI355X integrator hot path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c2i|c2iv|c2bcss|c3|c3b|c4|c5] [--traj-len L]

Contract (driver): W untimed warm-up passes, then EXACTLY K timed passes bracketed by a barrier +
device synchronise on both sides; the maximum over ranks is the job time; rank 0 prints ONE JSON
line.  A "step" of the bench is one pass of the hot path over one batch of synthetic input: one
call of the C-ABI entry point integrating a trajectory of L leapfrog steps for every chain of the
rank's shard (L = the trajectory length SURVEY.md section 8d quotes for the config).  `value` is
leapfrog steps (Integrator.step equivalents) per second summed over all chains and ranks, with the
inputs resident in HBM when the timed region starts; failed chains count only completed steps.

Default (N=1) workload = BASELINE.json configs[1] (c2): EuclideanMetricSystem, dense-precision
Gaussian target, D=128, 4096 chains per GPU, identity metric, explicit leapfrog h=0.05, L=1000.
Multi-GPU: chains are sharded (weak scaling: 4096 chains per GPU); the path has no exchange step, so the
timed region contains NO collective.  The one collective of a sampling job - the
RCCL all-gather of positions at trace collection - is timed once, separately, after the timed region and
reported as `config.trace_gather_ms`; MICI_AMD_BENCH_GATHER=rccl|gloo-host puts one gather per trajectory
inside the timed region instead (overlapped with the next trajectory).
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP64_MFMA_PEAK_TF = 78.6   # MI355X dense FP64 matrix peak (256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz)


def _make_spd(dim, rng):
    """P = A A^T / D + I with A ~ N(0,1)^{DxD}."""
    a = rng.standard_normal((dim, dim))
    return a @ a.T / dim + np.eye(dim)


def _torus_init(n, rng, R=1.0, r=0.5):
    """Initial positions on the torus from (theta, phi) ~ U(0, 2 pi) (reference README.md:344-355)."""
    theta, phi = rng.uniform(0, 2 * np.pi, size=(2, n))
    return np.stack([(R + r * np.cos(phi)) * np.cos(theta), (R + r * np.cos(phi)) * np.sin(theta),
                     r * np.sin(phi)], -1)


def make_workload(config, n_chains, rng):
    """Synthetic inputs of SURVEY.md section 8d.  Returns dict with the device system, integrator, initial
    state and the algorithmic work per chain-step.  The oracle twin (`make_oracle`) is only constructed by the
    cpu_baseline leg: nothing under oracle/ is imported on the measured path."""
    from mici_amd import integrators, models, systems

    if config in ("c2", "c2i", "c2iv", "c2bcss"):
        dim, h, traj = 128, 0.05, 1000
        stages = 3 if config == "c2bcss" else 1  # gradient evaluations per integrator step
        if config == "c2i":
            target, P = models.GaussIso(dim), None
            metric = None
            flops = 8.0 * dim
            name = "c2(i) iso-Gaussian"
        else:
            P = _make_spd(dim, rng)
            target = models.GaussDense(P)
            metric = P if config == "c2iv" else None
            flops = stages * (2.0 * dim * dim * (2 if config == "c2iv" else 1) + 8.0 * dim)
            name = "c2(iv) dense-Gaussian + dense metric" if config == "c2iv" else \
                "c2(iii) dense-precision Gaussian"
        system = systems.EuclideanMetricSystem(target, metric=metric)
        mk = 0 if metric is None else 2

        def make_oracle():
            from oracle import integrators as orc
            from oracle import models as omdl
            return orc.EuclidSystem(omdl.GaussIso(dim) if P is None else omdl.GaussDense(P), mk, metric)

        if config == "c2bcss":  # the three-stage BCSS composition on the c2(iii) workload
            integ = integrators.BCSSThreeStageIntegrator(system, h)
            iname = "BCSSThreeStageIntegrator (one step = 3 gradient evaluations)"
        else:
            integ = integrators.LeapfrogIntegrator(system, h)
            iname = "LeapfrogIntegrator"
        q0 = rng.standard_normal((n_chains, dim))
        z = rng.standard_normal((n_chains, dim))
        p0 = z if metric is None else z @ np.linalg.cholesky(metric).T
        return dict(name=f"{name}, EuclideanMetricSystem + {iname}", dim=dim, h=h,
                    coefficients=getattr(integ, "coefficients", None),
                    traj=traj, integ=integ, system=system, make_oracle=make_oracle, q0=q0, p0=p0,
                    bytes_per_chain_step=32.0 * dim, flops_per_chain_step=flops,
                    bound="hbm" if config == "c2i" else "mfma", kind="euclid")
    if config in ("c3", "c4"):
        dim, h, traj = (64, 0.02, 100) if config == "c3" else (256, 0.01, 50)
        base = _make_spd(dim, rng)
        system = systems.DenseRiemannianMetricSystem(models.Banana(dim), models.Rank1Metric(base))

        def make_oracle():
            from oracle import integrators as orc
            from oracle import models as omdl
            return orc.RiemannianSystem(omdl.Banana(dim), omdl.Rank1Metric(base))

        integ = integrators.ImplicitLeapfrogIntegrator(system, h)
        q0 = rng.standard_normal((n_chains, dim))
        p0 = system.sample_momentum_batch(q0, rng.standard_normal((n_chains, dim)))
        return dict(name=f"{config}(a) DenseRiemannianMetricSystem (rank-one-update dense metric, banana "
                         "target) + ImplicitLeapfrogIntegrator", dim=dim, h=h, traj=traj, integ=integ,
                    system=system, make_oracle=make_oracle, q0=q0, p0=p0, bytes_per_chain_step=32.0 * dim,
                    flops_per_chain_step=None, bound="mfma", kind="riemann")
    if config == "c3b":
        dim, h, traj = 64, 0.02, 100
        wts = np.linspace(0.5, 2.0, dim - 1)
        system = systems.SoftAbsRiemannianMetricSystem(models.Funnel(wts), softabs_coeff=1.0)

        def make_oracle():
            from oracle import integrators as orc
            from oracle import models as omdl
            return orc.RiemannianSystem(omdl.Funnel(wts), None, 1.0)

        integ = integrators.ImplicitLeapfrogIntegrator(system, h)
        q0 = rng.standard_normal((n_chains, dim))
        p0 = system.sample_momentum_batch(q0, rng.standard_normal((n_chains, dim)))
        return dict(name="c3(b) SoftAbsRiemannianMetricSystem (scaled funnel) + "
                         "ImplicitLeapfrogIntegrator", dim=dim, h=h, traj=traj, integ=integ,
                    system=system, make_oracle=make_oracle, q0=q0, p0=p0, bytes_per_chain_step=32.0 * dim,
                    flops_per_chain_step=None, bound="mfma", kind="softabs")
    if config == "c5":
        dim, h, traj = 3, 0.1, 1000
        system = systems.DenseConstrainedEuclideanMetricSystem(models.Torus(), models.TorusConstr())

        def make_oracle():
            from oracle import integrators as orc
            from oracle import models as omdl
            return orc.ConstrainedSystem(omdl.Torus(), omdl.TorusConstr())

        integ = integrators.ConstrainedLeapfrogIntegrator(system, h)
        q0 = _torus_init(n_chains, rng)
        p0 = system.sample_momentum_batch(q0, rng.standard_normal((n_chains, dim)))
        return dict(name="c5 DenseConstrainedEuclideanMetricSystem (README torus) + "
                         "ConstrainedLeapfrogIntegrator (Newton)", dim=dim, h=h, traj=traj, integ=integ,
                    system=system, make_oracle=make_oracle, q0=q0, p0=p0, bytes_per_chain_step=32.0 * dim,
                    flops_per_chain_step=1500.0, bound="hbm", kind="constrained")
    raise SystemExit(f"unknown --config {config}")


def cpu_baseline(w, budget_s=20.0):
    """The NumPy oracle (vectorised over chains, BLAS threads as configured) timed on a bounded
    sample of the same workload on this box's host cores."""
    from oracle import integrators as orc

    osys = w["make_oracle"]()
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        cores = os.cpu_count() or 1
    if w["kind"] != "euclid":
        # per-chain NumPy oracle (how the reference itself runs: one chain at a time, one core)
        fn = orc.constrained_leapfrog_steps if w["kind"] == "constrained" else orc.implicit_leapfrog_steps
        steps = {"riemann": 5, "softabs": 3, "constrained": 50}[w["kind"]]
        done, n1 = 0, 0
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < budget_s and n1 < w["q0"].shape[0]:
            _, _, _, nd = fn(osys, w["q0"][n1], w["p0"][n1], w["h"], steps)
            done += nd
            n1 += 1
        dt = time.perf_counter() - t0
        return dict(value=done / dt, unit="leapfrog-steps/s", cores=1, kind="port",
                    sample=f"oracle per-chain NumPy: {n1} chains x {steps} steps of the same workload "
                           f"in {dt:.1f} s (1 thread; BLAS single-threaded at these sizes)")
    n, steps = w["q0"].shape[0], 20
    coefs = w.get("coefficients")
    if coefs is not None:
        import functools
        orc_batch = functools.partial(orc.leapfrog_steps_batch, coefficients=list(coefs))
        orc_single = lambda s_, q_, p_, h_, n_: orc.composition_steps(  # noqa: E731
            s_, q_, p_, h_, n_, list(coefs)[:(len(coefs) - 3) // 2])
    else:
        orc_batch, orc_single = orc.leapfrog_steps_batch, orc.leapfrog_steps
    t0 = time.perf_counter()
    orc_batch(osys, w["q0"], w["p0"], w["h"], steps)
    dt = time.perf_counter() - t0
    # scale the sample to ~budget_s of CPU work, capped at the real trajectory length
    steps2 = int(min(w["traj"], max(steps, steps * budget_s / max(dt, 1e-6))))
    t0 = time.perf_counter()
    orc_batch(osys, w["q0"], w["p0"], w["h"], steps2)
    dt = time.perf_counter() - t0
    value = n * steps2 / dt
    # reference-style figure: one chain at a time on one core
    t0 = time.perf_counter()
    n1 = 0
    while time.perf_counter() - t0 < 2.0:
        orc_single(osys, w["q0"][n1 % n], w["p0"][n1 % n], w["h"], 100)
        n1 += 1
    single = n1 * 100 / (time.perf_counter() - t0)
    return dict(value=value, unit="leapfrog-steps/s", cores=int(cores), kind="port",
                sample=f"oracle.leapfrog_steps_batch (NumPy, vectorised over chains): {n} chains x "
                       f"{steps2} steps of the same workload in {dt:.1f} s",
                single_chain_1core=single)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c2")
    ap.add_argument("--chains-per-gpu", type=int, default=None)
    ap.add_argument("--traj-len", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    dist = None
    if world > 1:
        import torch  # plumbing only: rendezvous, barrier, max-over-ranks (CPU tensors over gloo)
        import torch.distributed as dist

        dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    from mici_amd import _ffi
    from mici_amd.runtime import Context, DeviceBatch

    ctx = Context(local_rank)
    n_local = args.chains_per_gpu or {"c3": 1024, "c3b": 1024, "c4": 1024, "c5": 2048}.get(args.config, 4096)
    rng = np.random.default_rng(1234 + rank)
    w = make_workload(args.config, n_local, rng)
    traj = args.traj_len or w["traj"]
    integ = w["integ"]

    batch = DeviceBatch(ctx, n_local, w["dim"])
    dirs = np.ones(n_local, dtype=np.int8)

    # optional RCCL communicator for the per-trajectory trace gather
    comm = None
    gather_mode = "none"
    pos_all = None
    if world > 1:
        import ctypes as C
        import torch

        gather_mode = os.environ.get("MICI_AMD_BENCH_GATHER", "off")
        in_loop = gather_mode in ("rccl", "gloo-host")

        def setup_comm():
            idbuf = torch.zeros(_ffi.MM_COMM_ID_BYTES, dtype=torch.uint8)
            if rank == 0:
                raw = (C.c_uint8 * _ffi.MM_COMM_ID_BYTES)()
                _ffi.check(ctx._lib.mm_comm_unique_id(raw), None, "mm_comm_unique_id")
                idbuf = torch.tensor(list(raw), dtype=torch.uint8)
            dist.broadcast(idbuf, src=0)
            raw = (C.c_uint8 * _ffi.MM_COMM_ID_BYTES)(*idbuf.tolist())
            h = C.c_void_p()
            _ffi.check(ctx._lib.mm_comm_create(ctx.handle, world, rank, raw, C.byref(h)),
                       ctx.handle, "mm_comm_create")
            return h, np.empty((world * n_local, w["dim"]))

        if gather_mode == "rccl":
            try:
                comm, pos_all = setup_comm()
            except Exception as e:  # keep the scaling run alive, but say so in the JSON line
                print(f"[bench] RCCL gather unavailable ({e}); falling back to host gather",
                      file=sys.stderr)
                gather_mode = "gloo-host"

    want_host = 1 if rank == 0 else 0  # only the trace-writing rank needs the gathered array on the host

    def collect_traces():
        """Trace collection, once per trajectory: RCCL all-gather over xGMI on the communicator's own
        stream (overlapped with the next trajectory), pinned host copy on rank 0 only."""
        if comm is not None:
            _ffi.check(ctx._lib.mm_comm_allgather_pos_async(comm, batch.handle, want_host),
                       ctx.handle, "mm_comm_allgather_pos_async")
        elif gather_mode == "gloo-host":
            import torch
            q, _, _ = batch.download()
            out = [torch.empty_like(torch.from_numpy(q)) for _ in range(world)]
            dist.all_gather(out, torch.from_numpy(q))

    def finish_traces():
        if comm is not None:
            _ffi.check(ctx._lib.mm_comm_wait(
                comm, pos_all.ctypes.data_as(_ffi.c_double_p) if want_host else None),
                ctx.handle, "mm_comm_wait")

    in_loop = world > 1 and gather_mode in ("rccl", "gloo-host")

    def one_pass():
        integ.step_device(batch, traj, ctx)
        if in_loop:
            collect_traces()

    def barrier():
        ctx.sync()
        if dist is not None:
            dist.barrier()

    batch.upload(w["q0"], w["p0"], dirs)
    for _ in range(args.warmup):
        one_pass()
    if in_loop:
        finish_traces()
    batch.upload(w["q0"], w["p0"], dirs)  # timed region starts from the same resident state

    barrier()
    t0 = time.perf_counter()
    kernel_ms = 0.0
    done_acc, counters_acc = 0.0, {}
    # Kernel time from HIP events on the stream the kernel runs on.  Explicit integrators: ONE pair around the
    # K back-to-back launches (events between launches were measured to open host-side gaps: 3.9 ms kernels
    # showing up as 6 ms passes).  Implicit / constrained integrators: a pair per launch (the status download of
    # every pass synchronises anyway); eight pairs are cycled and read back only when reused.
    n_pairs = 8
    per_launch_events = w["kind"] != "euclid"
    if not per_launch_events:
        ctx.record(0)
    for k in range(args.steps):
        s = (k % n_pairs) * 2
        if per_launch_events:
            if k >= n_pairs:
                kernel_ms += ctx.elapsed_ms(s, s + 1)
            ctx.record(s)
        integ.step_device(batch, traj, ctx)
        if per_launch_events:
            ctx.record(s + 1)
        if in_loop:
            collect_traces()  # trace collection once per trajectory
        if w["kind"] != "euclid":
            _, nd = batch.download_status()  # the sampler needs this per trajectory anyway
            done_acc += float(nd.sum())
            for key, val in (integ.last_counters or {}).items():
                counters_acc[key] = counters_acc.get(key, 0) + val
    if not per_launch_events:
        ctx.record(1)
    if in_loop:
        finish_traces()  # the last gather must have landed inside the timed region
    barrier()
    elapsed = time.perf_counter() - t0
    if per_launch_events:
        for k in range(max(0, args.steps - n_pairs), args.steps):  # the launches whose events were not read yet
            s = (k % n_pairs) * 2
            kernel_ms += ctx.elapsed_ms(s, s + 1)
    else:
        kernel_ms = ctx.elapsed_ms(0, 1)  # K launches back to back
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    done_local = float(n_local) * traj * args.steps if w["kind"] == "euclid" else done_acc
    total_steps = done_local
    if dist is not None:
        import torch
        t = torch.tensor([done_local], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        total_steps = float(t.item())

    # The job's one collective, timed on its own AFTER the timed region (communicator creation included in
    # neither).  Guarded by a timeout: a wedged RCCL bootstrap must not cost the scaling run its result line.
    trace_gather_ms = None
    exit_hard = False
    if world > 1 and not in_loop:
        import threading
        box = {}

        def timed_gather():
            nonlocal comm, pos_all
            try:
                comm, pos_all = setup_comm()
                for rep in range(2):  # first gather warms the rings up
                    barrier()
                    tg = time.perf_counter()
                    collect_traces()
                    finish_traces()
                    barrier()
                    box["ms"] = (time.perf_counter() - tg) * 1e3
                box["mode"] = "rccl, outside the timed region"
            except Exception as e:
                box["mode"] = f"rccl unavailable ({type(e).__name__}: {str(e)[:160]})"

        th = threading.Thread(target=timed_gather, daemon=True)
        th.start()
        th.join(timeout=90.0)
        if th.is_alive():
            gather_mode, exit_hard = "rccl bootstrap timed out", True
        else:
            gather_mode, trace_gather_ms = box.get("mode", "?"), box.get("ms")

    if rank == 0:
        value = total_steps / elapsed
        launch_s = kernel_ms / 1e3 / args.steps
        chain_steps_per_launch = n_local * traj
        if w["kind"] == "softabs":
            # algorithmic flops: n_eig * 9 D^3 (symmetric eigendecomposition
            # with vectors) + 4 D^3 per momentum-solve evaluation (grad_quadratic_form_inv's two GEMMs)
            d = float(w["dim"])
            n_m = counters_acc.get("n_metric", 0)
            n_b = max(counters_acc.get("n_fp_evals", 0) - n_m, 0)
            flops_total = n_m * 9 * d**3 + n_b * 4 * d**3
            w["flops_per_chain_step"] = flops_total / max(done_local, 1.0)
            chain_steps_per_launch = done_local / args.steps
        if w["kind"] == "riemann":
            # algorithmic flops of SURVEY.md section 8d from the device work counters:
            #   n_M D^3/3 (factorisations) + n_inv 2D^3/3 (one explicit inverse per completed step)
            #   + (2 n_M + 3 n_B) D^2 (solves / mat-vecs / outer products), n_B = momentum-solve evals
            d = float(w["dim"])
            n_m = counters_acc.get("n_metric", 0)
            n_b = max(counters_acc.get("n_fp_evals", 0) - n_m, 0)
            flops_total = n_m * d**3 / 3 + done_local * 2 * d**3 / 3 + (2 * n_m + 3 * n_b) * d * d
            w["flops_per_chain_step"] = flops_total / max(done_local, 1.0)
            chain_steps_per_launch = done_local / args.steps
        if w["bound"] == "mfma":
            achieved = w["flops_per_chain_step"] * chain_steps_per_launch / launch_s / 1e12
            roof = dict(bound="mfma", achieved=achieved, peak=FP64_MFMA_PEAK_TF, unit="TFLOP/s",
                        frac=achieved / FP64_MFMA_PEAK_TF, traffic=None)
        else:
            # a launch integrates the whole trajectory with the chain state in registers: the algorithmic HBM
            # traffic is one read and one write of (pos, mom) per chain per LAUNCH, not per step
            achieved = w["bytes_per_chain_step"] * n_local / launch_s / 1e9
            roof = dict(bound="hbm", achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=achieved / HBM_PEAK_GBS, traffic=None)
            if w.get("flops_per_chain_step"):  # what actually limits these kernels: FP64 vector issue
                tf = w["flops_per_chain_step"] * chain_steps_per_launch / launch_s / 1e12
                roof["fp64_valu"] = dict(achieved=tf, peak=FP64_MFMA_PEAK_TF, unit="TFLOP/s",
                                         frac=tf / FP64_MFMA_PEAK_TF)
        # HBM traffic per launch from the committed rocprofv3 PMC passes of this same command
        # (FETCH_SIZE / WRITE_SIZE cannot be read from inside the process); null when not profiled
        pmc_name = "r01_c2_pmc_hbm.json" if args.config == "c2" else f"r01b_{args.config}_pmc_hbm.json"
        pmc = os.path.join(ROOT, "profiles", pmc_name)
        default_shape = args.chains_per_gpu is None and args.traj_len is None
        if default_shape and os.path.exists(pmc):
            with open(pmc) as fh:
                roof["traffic"] = json.load(fh)["traffic_bytes_per_launch"]
            roof["traffic_source"] = f"profiles/{pmc_name} (rocprofv3 --pmc, corrected)"
        roof["kernel_ms_per_launch"] = kernel_ms / args.steps
        roof["algorithmic_flops_per_chain_step"] = w["flops_per_chain_step"]
        roof["algorithmic_bytes_per_chain_step"] = w["bytes_per_chain_step"] / (traj if w["bound"] == "hbm" else 1)
        if counters_acc:
            roof["work_counters"] = counters_acc
        out = {
            "metric": "leapfrog-steps/sec (all chains)",
            "value": value,
            "unit": "leapfrog-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{w['name']}, D={w['dim']}, {n_local} chains/GPU x {world} GPU, "
                            f"h={w['h']}, one pass = a trajectory of {traj} leapfrog steps per chain",
                "baseline_config": {"c2": "BASELINE.json configs[1]", "c3": "BASELINE.json configs[2] (Cholesky path)",
                                    "c3b": "BASELINE.json configs[2] (SoftAbs path)",
                                    "c4": "BASELINE.json configs[3] (per-GPU shard)",
                                    "c5": "BASELINE.json configs[4] (per-GPU shard)"}.get(args.config, args.config),
                "chains_per_gpu": n_local, "dim": w["dim"], "traj_len": traj,
                "parallelism": f"chains sharded x{world}, no collective in the timed region"
                               if not in_loop else f"chains sharded x{world}, trace gather per pass: {gather_mode}",
                "trace_gather": gather_mode if world > 1 else "none",
                "trace_gather_ms": trace_gather_ms,
            },
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(w)
        print(json.dumps(out), flush=True)

    if exit_hard:  # a collective is wedged in the helper thread: the result line is out, leave without cleanup
        sys.stdout.flush()
        os._exit(0)
    if comm is not None:
        ctx._lib.mm_comm_destroy(comm)
    batch.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
