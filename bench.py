#!/usr/bin/env python3
"""bench.py - leapfrog-steps/sec (all chains) of the MI355X integrator hot path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c2i|c2i_stream|c2iv|c2bcss|c3|c3b|c4|c5|c3_user|c4_general|c4_user_lowrank|c3b_dense|c4_d512|c3b_d128|c3b_d256]
                    [--traj-len L] [--chains-per-gpu M] [--no-extra-configs] [--no-cpu-baseline]

Contract (driver): W untimed warm-up passes, then EXACTLY K timed passes bracketed by a barrier +
device synchronise on both sides; the maximum over ranks is the job time; rank 0 prints ONE JSON
line.  A "step" of the bench is one pass of the hot path over one batch of synthetic input: one
call of the C-ABI entry point integrating a trajectory of L leapfrog steps for every chain of the
rank's shard (L = the trajectory length SURVEY.md section 8d quotes for the config).  `value` is
leapfrog steps (Integrator.step equivalents) per second summed over all chains and ranks, with the
inputs resident in HBM when the timed region starts; failed chains count only completed steps.

Headline (N=1) workload = BASELINE.json configs[1] (c2): EuclideanMetricSystem, dense-precision
Gaussian target, D=128, 4096 chains per GPU, identity metric, explicit leapfrog h=0.05, L=1000.
The other BASELINE configs are measured in the same process by the same procedure (c2(i), c2(iv), c3(a), c3(b), the c4
per-GPU shard, the c5 per-GPU shard, the three user-source configs): their full records (value, ms_per_step, roofline,
work counters and, at N=1, cpu_baseline each) go to the SIDECAR file `bench_configs.json` next to this script; the one
stdout line stays under 4 KB (`compact_result`: the headline whole, five scalars per extra config) because the driver's
capture lost the 22 KB line of round 4.  Their pass counts are capped so that the default run stays within a few minutes
(`steps` is reported per entry in the sidecar).

Multi-GPU: one process per GPU.  Launched under `python -m torch.distributed.run` the ranks come from
RANK / LOCAL_RANK / WORLD_SIZE; started plainly as `python bench.py --gpus N` the script starts the N
rank processes itself and fails if fewer than N devices are visible.  The ranks rendezvous over a
Unix-domain socket (mici_amd/rendezvous.py, standard library only - no torch anywhere in this file).
Chains are sharded (weak scaling: the per-GPU shard is fixed); the path has no exchange step, so the
timed region contains NO collective.  The one collective of a sampling job - the
RCCL all-gather of positions over xGMI at trace collection - is timed once, separately, after the timed
region of the headline config and reported as `config.trace_gather_ms`; MICI_AMD_BENCH_GATHER=rccl puts
one gather per trajectory inside the timed region instead (overlapped with the next trajectory).

MICI_AMD_SHARE_DEVICE=1 (a TEST switch, tests/test_gpu_multirank.py): the N ranks share the visible devices round-robin
instead of refusing to oversubscribe, and the trace is gathered through the host rendezvous (RCCL refuses two ranks on one
device) - N > 1 rank processes with their own contexts on a one-GPU box; the numbers of such a run are not a measurement.
MICI_AMD_BENCH_DUMP_TRACE=<file.npy>: rank 0 writes the gathered positions of the headline config there.

Re-timing: a single-process run whose timed region's wall clock exceeds the HIP-event kernel time of the same passes by
more than 10 % (the device queue was descheduled mid-region: DESIGN.md section 8) times the region again, at most twice
more, and reports the fastest attempt; `roofline.attempts` lists all of them.
"""

from __future__ import annotations

import argparse
import json
import os
import subprocess
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


class _OracleMomenta:
    """Initial momenta for the CPU-baseline worker (no device there): p = M(q)^{1/2} z through the oracle, filled
    in for the chain ranges the bounded CPU sample actually touches (`fix(lo, hi)`, outside the timed loops)."""

    def __init__(self, kind, osys, q0, z):
        self.kind, self.osys, self.q0, self.z = kind, osys, q0, z
        self.p0 = np.array(z, copy=True)
        self.done = np.zeros(q0.shape[0], dtype=bool)

    def fix(self, lo, hi):
        from oracle import integrators as orc

        for c in range(lo, min(hi, self.q0.shape[0])):
            if self.done[c]:
                continue
            if self.kind == "constrained":
                self.p0[c] = self.osys.project_onto_cotangent_space(
                    self.osys.msqrt(self.z[c]), self.osys.constraint.jacob_constr(self.q0[c]))
            else:
                self.p0[c] = self.osys.sample_momentum(orc._State(self.q0[c], self.z[c]), self.z[c])
            self.done[c] = True


def _oracle_momenta(kind, osys, q0, z):
    return _OracleMomenta(kind, osys, q0, z)


def make_workload(config, n_chains, rng, device=True, chain_rng=None):
    """Synthetic inputs of SURVEY.md section 8d.  Returns dict with the device system, integrator, initial
    state and the algorithmic work per chain-step.  `rng` draws the MODEL (precision / base matrices: the same on
    every rank - DESIGN section 6, model parameters are replicated) and, unless `chain_rng` is given, the chains'
    initial states after it; ranks > 0 pass their own `chain_rng`.  The oracle twin (`make_oracle`) is only constructed by the
    cpu_baseline leg (`device=False`: its own interpreter, initial momenta through the oracle, no device objects
    touched): nothing under oracle/ is imported on the measured path."""
    from mici_amd import integrators, models, systems

    crng = chain_rng if chain_rng is not None else rng  # drawn from AFTER the model, as rank 0 always did

    if config in ("c2", "c2i", "c2iv", "c2bcss", "c2i_stream"):
        dim, h, traj = 128, 0.05, 1000
        if config == "c2i_stream":
            # the HBM-bound regime north_star names ("coalesced HBM loads of the per-chain (q, p) state"): ONE leapfrog step per
            # launch, so a launch is one read and one write of the state - 32 D bytes per chain-step, 4.3 GB at 2^20 chains -
            # and nothing else (integrators.py:170-173, systems.py:143-152, 362-363)
            traj = 1
        stages = 3 if config == "c2bcss" else 1  # gradient evaluations per integrator step
        if config in ("c2i", "c2i_stream"):
            target, P = models.GaussIso(dim), None
            metric = None
            flops = 8.0 * dim
            # what leapfrog_elem_kernel executes per chain-step and coordinate for this target (grad = q): three
            # v_fma_f64 - p -= (t/2) g, q += t p, p -= (t/2) g - i.e. 6 flops against SURVEY 8d's ~8
            exec_flops = 6.0 * dim
            name = "c2(i) iso-Gaussian" if config == "c2i" else "c2(i) iso-Gaussian, streaming regime (one step per launch)"
        else:
            P = _make_spd(dim, rng)
            target = models.GaussDense(P)
            metric = P if config == "c2iv" else None
            flops = stages * (2.0 * dim * dim * (2 if config == "c2iv" else 1) + 8.0 * dim)
            exec_flops = None
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
        q0 = crng.standard_normal((n_chains, dim))
        z = crng.standard_normal((n_chains, dim))
        p0 = z if metric is None else z @ np.linalg.cholesky(metric).T
        return dict(name=f"{name}, EuclideanMetricSystem + {iname}", dim=dim, h=h,
                    coefficients=getattr(integ, "coefficients", None),
                    traj=traj, integ=integ, system=system, make_oracle=make_oracle, q0=q0, p0=p0,
                    bytes_per_chain_step=32.0 * dim, flops_per_chain_step=flops, valu_executed_flops_per_chain_step=exec_flops,
                    bound="hbm" if config in ("c2i", "c2i_stream") else "mfma", kind="euclid")
    if config in ("c3", "c4", "c3_user", "c4_general", "c4_user_lowrank", "c4_d512"):
        # c3_user / c4_general (VERDICT r03 #1c): the GENERAL dense-Riemannian path - the metric reaches the library as user
        # source (mici_amd/user_examples.py), compiled around the matrix-core kernels at run time.  c3_user: a metric that
        # is not built in (softplus diagonal + rank one, D = 64); c4_general: the c4 workload itself with its rank-one
        # metric handed over as user source (D = 256) - M(x) v of the refinement solves from the user's entries.
        # c4_d512 (round 5; round 6: c4's own trajectory length, 50 - it was 5 while a pass took 0.1 s a step): the c4 workload
        # at twice the dimension - beyond what a CU's registers hold, on the global-memory
        # tier (csrc/implicit_global.h: the chain's metric in HBM, blocked sweep, column-walk products)
        dim, h, traj = (64, 0.02, 100) if config in ("c3", "c3_user") else ((512, 0.008, 50) if config == "c4_d512"
                                                                              else (256, 0.01, 50))
        if config == "c3_user":
            from mici_amd import user_examples
            cvec = 0.5 * rng.standard_normal(dim)
            rmetric = models.UserMetric(dim, user_examples.SOFTPLUS_RANK1_FAST, cvec)
            mname = "softplus-diagonal + rank-one metric as USER SOURCE (hipRTC, MM_USER_AUX + MM_USER_VJP_FLAT)"
        else:
            base = _make_spd(dim, rng)
            if config == "c4_general":
                from mici_amd import user_examples
                rmetric = models.UserMetric(dim, user_examples.RANK1_AS_USER_FLAT, base)
                mname = "rank-one-update dense metric as USER SOURCE (hipRTC, MM_USER_VJP_FLAT)"
            elif config == "c4_user_lowrank":
                # round 6: the same user source DECLARING its structure (user_metric.h MM_USER_LOWRANK: M = C + s u(q) u(q)^T) - the
                # Woodbury path of DESIGN section 4.3f around a user metric; c4_general stays the undeclared, generic path
                from mici_amd import user_examples
                rmetric = models.UserMetric(dim, user_examples.RANK1_AS_USER_LOWRANK, base)
                mname = "rank-one-update dense metric as USER SOURCE that declares its structure (hipRTC, MM_USER_VJP_FLAT + MM_USER_LOWRANK)"
            else:
                rmetric = models.Rank1Metric(base)
                mname = "rank-one-update dense metric"
        system = systems.DenseRiemannianMetricSystem(models.Banana(dim), rmetric)

        def make_oracle():
            from oracle import integrators as orc
            from oracle import models as omdl
            return orc.RiemannianSystem(omdl.Banana(dim), omdl.SoftPlusRank1Metric(cvec) if config == "c3_user"
                                        else omdl.Rank1Metric(base))

        integ = integrators.ImplicitLeapfrogIntegrator(system, h)
        q0 = crng.standard_normal((n_chains, dim))
        z = crng.standard_normal((n_chains, dim))
        p0 = system.sample_momentum_batch(q0, z) if device else _oracle_momenta("riemann", make_oracle(), q0, z)
        mom = None if device else p0
        p0 = p0 if device else mom.p0
        return dict(name=f"{config if '_' in config else config + '(a)'} DenseRiemannianMetricSystem ({mname}, banana "
                         "target) + ImplicitLeapfrogIntegrator", dim=dim, h=h, traj=traj, integ=integ,
                    system=system, make_oracle=make_oracle, q0=q0, p0=p0, momenta=mom, bytes_per_chain_step=32.0 * dim,
                    flops_per_chain_step=None, bound="mfma", kind="riemann")
    if config in ("c3b", "c3b_dense", "c3b_d128", "c3b_d256"):
        # c3b_dense (VERDICT r03 #1c): SoftAbs on the BANANA - its tridiagonal Hessian and matrix-Tressian product reach the
        # library as user source (mici_amd/user_examples.py BANANA_HESS): the dense path - G = A X on the matrix cores,
        # grad_log_abs_det / grad_quadratic_form_inv formed in full - none of the arrowhead structure c3(b) leans on
        dim, h, traj = 64, 0.02, 100
        wide = config in ("c3b_d128", "c3b_d256")
        if wide:
            # round 6: the SoftAbs workspace tiers - 64 < D <= 256, the three matrices of a chain in a global-memory
            # workspace (csrc/softabs.h NP = 128 / 256) - on the c3(b) funnel at two and four times its dimension
            # c3(b)'s own trajectory length (100; until the end of round 6 these entries ran 20 / 10 steps, where the cold first
            # decomposition of a pass - it restarts from q0 - weighed 5 - 10 times as much: D = 256 3.0e3 steps/s at 10 steps,
            # 5.1e3 at 40)
            dim, traj = (128, 100) if config == "c3b_d128" else (256, 100)
        wts = np.linspace(0.5, 2.0, dim - 1)
        if config == "c3b_dense":
            h = 0.01  # (at 0.02 half of the banana chains meet a ConvergenceError within 100 steps - in the reference too)
            from mici_amd import user_examples
            system = systems.SoftAbsRiemannianMetricSystem(
                models.Banana(dim), softabs_coeff=1.0, hess_neg_log_dens=models.UserHessian(user_examples.BANANA_HESS))
        else:
            system = systems.SoftAbsRiemannianMetricSystem(models.Funnel(wts), softabs_coeff=1.0)

        def make_oracle():
            from oracle import integrators as orc
            from oracle import models as omdl
            return orc.RiemannianSystem(omdl.Banana(dim) if config == "c3b_dense" else omdl.Funnel(wts), None, 1.0)

        integ = integrators.ImplicitLeapfrogIntegrator(system, h)
        q0 = (0.5 if wide else 1.0) * crng.standard_normal((n_chains, dim))
        z = crng.standard_normal((n_chains, dim))
        p0 = system.sample_momentum_batch(q0, z) if device else _oracle_momenta("softabs", make_oracle(), q0, z)
        mom = None if device else p0
        p0 = p0 if device else mom.p0
        return dict(name=("c3b_dense SoftAbsRiemannianMetricSystem (banana target, Hessian / MTP as USER SOURCE: dense path) + "
                          if config == "c3b_dense" else
                          f"{config} SoftAbsRiemannianMetricSystem (scaled funnel, workspace tier) + " if wide else
                          "c3(b) SoftAbsRiemannianMetricSystem (scaled funnel) + ")
                    + "ImplicitLeapfrogIntegrator", dim=dim, h=h, traj=traj, integ=integ,
                    system=system, make_oracle=make_oracle, q0=q0, p0=p0, momenta=mom, bytes_per_chain_step=32.0 * dim,
                    flops_per_chain_step=None, bound="mfma", kind="softabs")
    if config == "c5":
        dim, h, traj = 3, 0.1, 1000
        system = systems.DenseConstrainedEuclideanMetricSystem(models.Torus(), models.TorusConstr())

        def make_oracle():
            from oracle import integrators as orc
            from oracle import models as omdl
            return orc.ConstrainedSystem(omdl.Torus(), omdl.TorusConstr())

        integ = integrators.ConstrainedLeapfrogIntegrator(system, h)
        q0 = _torus_init(n_chains, crng)
        z = crng.standard_normal((n_chains, dim))
        p0 = system.sample_momentum_batch(q0, z) if device else _oracle_momenta("constrained", make_oracle(), q0, z)
        mom = None if device else p0
        p0 = p0 if device else mom.p0
        return dict(name="c5 DenseConstrainedEuclideanMetricSystem (README torus) + "
                         "ConstrainedLeapfrogIntegrator (Newton)", dim=dim, h=h, traj=traj, integ=integ,
                    system=system, make_oracle=make_oracle, q0=q0, p0=p0, momenta=mom, bytes_per_chain_step=32.0 * dim,
                    flops_per_chain_step=1500.0, bound="hbm", kind="constrained")
    raise SystemExit(f"unknown --config {config}")



def _sweep_mfma_counts(dim):
    """v_mfma_f64_16x16x4 instructions per metric factorisation of the dense-Riemannian kernels (full sweep = explicit
    inverse, solve = trailing LDL^T sweep), counted from the kernel sources; None for the VALU kernels."""
    if 32 < dim <= 64:      # k_implicit_mfma.hip: 16 blocks of 4 pivots, 10 lower tiles; trailing: tiles with J >= I0
        return dict(full=160, solve=4 * (10 + 6 + 3 + 1), padded_dim=64)
    if dim > 279:           # implicit_global.h (round 6): per NB-pivot block NB / 4 MFMAs per lower tile + (NB / 16)(NB / 4) per tile
        dp = (dim + 63) & ~63  # row for the W operands; products are column walks (no matrix-core instructions)
        nb = 32 if dp <= 512 else 16
        nt = (dim + 15) // 16
        per_block = nt * (nt + 1) // 2 * (nb // 4) + nt * (nb // 16) * (nb // 4)
        return dict(full=-(-dim // nb) * per_block, solve=0, padded_dim=dp)
    if 75 < dim <= 256:     # k_implicit_blk16.hip: per 16-pivot block 4 MFMAs per updated tile + 64 (-W) + 4 (pivot block)
        nblk = (dim + 15) // 16
        full = nblk * (136 * 4 + 64 + 4)
        solve = sum((16 - b) * (17 - b) // 2 * 4 + 64 + 4 for b in range(nblk))
        return dict(full=full, solve=solve, padded_dim=256)
    return None


DEFAULT_CHAINS = {"c3": 1024, "c3b": 1024, "c4": 1024, "c5": 2048, "c3_user": 1024, "c4_general": 1024, "c4_user_lowrank": 1024, "c3b_dense": 1024,
                  "c4_d512": 256, "c2i_stream": 1 << 20, "c3b_d128": 256, "c3b_d256": 256}  # per GPU; else 4096
CPU_CHAINS = {"c2i_stream": 1 << 16}  # chains of the cpu_baseline sample where the device shard would not fit a host's pool
EXTRA_CONFIGS = ("c2i", "c2i_stream", "c2iv", "c3", "c3b", "c4", "c5", "c3_user", "c4_general", "c4_user_lowrank", "c3b_dense", "c4_d512",
                 "c3b_d128", "c3b_d256")
# pass counts of the extra configs are capped (a c3(b) pass is ~1 s, a c4 pass ~0.1-0.3 s)
EXTRA_STEP_CAP = {"c3b": 5, "c4": 10, "c4_general": 10, "c4_user_lowrank": 10, "c3b_dense": 5, "c4_d512": 3, "c3b_d128": 3, "c3b_d256": 2}
BASELINE_CONFIG = {"c2": "BASELINE.json configs[1]", "c2i": "BASELINE.json configs[1] (iso-Gaussian variant, SURVEY 8d c2(i))",
                   "c2i_stream": "BASELINE.json configs[1] sizes (iso-Gaussian, D = 128) with n_steps = 1 and 2^20 chains: the "
                                 "HBM-bound regime of the (q, p) state loads north_star names",
                   "c2iv": "BASELINE.json configs[1] (dense-metric variant, SURVEY 8d c2(iv))",
                   "c3": "BASELINE.json configs[2] (Cholesky path)", "c3b": "BASELINE.json configs[2] (SoftAbs path)",
                   "c4": "BASELINE.json configs[3] (per-GPU shard)", "c5": "BASELINE.json configs[4] (per-GPU shard)",
                   "c3_user": "BASELINE.json configs[2] sizes, a metric_func that is not built in (user source)",
                   "c4_general": "BASELINE.json configs[3] (per-GPU shard), its metric_func handed over as user source",
                   "c4_user_lowrank": "BASELINE.json configs[3] (per-GPU shard), its metric_func handed over as user source that declares its constant + rank-one structure",
                   "c3b_dense": "BASELINE.json configs[2] (SoftAbs path) on the banana target: a dense (user-source) Hessian",
                   "c4_d512": "BASELINE.json configs[3] at twice the dimension (D = 512: the global-memory tier)",
                   "c3b_d128": "BASELINE.json configs[2] (SoftAbs path) at D = 128, 256 chains: the workspace tier NP = 128",
                   "c3b_d256": "BASELINE.json configs[2] (SoftAbs path) at D = 256, 256 chains: the workspace tier NP = 256"}


# ---- CPU baseline: the oracle on this box's host cores ------------
def _cpu_noop(_):
    return 0


def _host_cores():
    """Cores this process may really use: the affinity mask, cut down to the cgroup CPU quota if there is one (a
    container can show 256 schedulable CPUs and be throttled to a handful)."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()[:2]
        if quota != "max":
            cores = max(1, min(cores, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return cores


def _cpu_shard_euclid(job):
    """Pool worker: vectorised-over-chains oracle on one shard of chains, repeated until the common deadline."""
    osys, q, p, h, steps, coefs, deadline = job
    from oracle import integrators as orc
    done, t0 = 0, time.perf_counter()
    while True:
        if coefs is not None:
            orc.leapfrog_steps_batch(osys, q, p, h, steps, coefficients=list(coefs))
        else:
            orc.leapfrog_steps_batch(osys, q, p, h, steps)
        done += q.shape[0] * steps
        if time.time() >= deadline:
            break
    return done, time.perf_counter() - t0


def _cpu_shard_chains(job):
    """Pool worker: per-chain oracle (how the reference itself runs), chain after chain until the common deadline.
    The worker prepares the initial momenta of its own chains (p = M(q)^{1/2} z) outside the timed part."""
    osys, kind, q, z, h, steps, deadline = job
    from oracle import integrators as orc
    fn = orc.constrained_leapfrog_steps if kind == "constrained" else orc.implicit_leapfrog_steps
    mom = _OracleMomenta(kind, osys, q, z)
    done, spent, c = 0, 0.0, 0
    while True:
        mom.fix(c % q.shape[0], c % q.shape[0] + 1)
        t0 = time.perf_counter()
        done += fn(osys, q[c % q.shape[0]], mom.p0[c % q.shape[0]], h, steps)[3]
        spent += time.perf_counter() - t0
        c += 1
        if time.time() >= deadline:
            break
    return done, spent


def cpu_baseline_measure(config, budget_s):
    """Runs in its own interpreter (see cpu_baseline): 1 BLAS thread per process, no HIP anywhere.
    (a) reference-style: one chain at a time on one core; (c) sharded over all host cores with a process pool whose
    workers run until a common deadline (the wall time is bounded whatever the box's real parallelism is)."""
    import multiprocessing as mp

    from oracle import integrators as orc

    cores = _host_cores()
    n = CPU_CHAINS.get(config, DEFAULT_CHAINS.get(config, 4096))
    w = make_workload(config, n, np.random.default_rng(1234), device=False)
    osys = w["make_oracle"]()
    coefs = w.get("coefficients")
    out = dict(unit="leapfrog-steps/s", cores=cores, kind="port")
    pool_s = budget_s / 2
    if w["kind"] == "euclid":
        if coefs is not None:
            free = list(coefs)[:(len(coefs) - 3) // 2]
            single_fn = lambda q_, p_, n_: orc.composition_steps(osys, q_, p_, w["h"], n_, free)  # noqa: E731
        else:
            single_fn = lambda q_, p_, n_: orc.leapfrog_steps(osys, q_, p_, w["h"], n_)  # noqa: E731
        t0, n1 = time.perf_counter(), 0
        while time.perf_counter() - t0 < min(2.0, budget_s / 4):
            single_fn(w["q0"][n1 % n], w["p0"][n1 % n], 100)
            n1 += 1
        single = n1 * 100 / (time.perf_counter() - t0)
        shard = max(1, n // cores)
        nw = min(cores, n // shard)
        mk = lambda r, dl: (osys, w["q0"][r * shard:(r + 1) * shard], w["p0"][r * shard:(r + 1) * shard], w["h"], 50,  # noqa: E731
                            coefs, dl)
        sample = f"oracle.leapfrog_steps_batch (NumPy, vectorised over chains): {nw} workers x {shard} chains"
        worker = _cpu_shard_euclid
    else:
        steps = {"riemann": 2 if w["dim"] > 128 else 5, "softabs": 3 if w["dim"] <= 64 else 1, "constrained": 50}[w["kind"]]
        fn = orc.constrained_leapfrog_steps if w["kind"] == "constrained" else orc.implicit_leapfrog_steps
        n_single = min(n // 2, 64)
        t0, n1, done, spent = time.perf_counter(), 0, 0, 0.0
        while time.perf_counter() - t0 < budget_s / 4 and n1 < n_single:
            w["momenta"].fix(n1, n1 + 1)
            t1 = time.perf_counter()
            done += fn(osys, w["q0"][n1], w["p0"][n1], w["h"], steps)[3]
            spent += time.perf_counter() - t1
            n1 += 1
        single = done / spent
        k = max(1, (n - n1) // cores)
        nw = min(cores, (n - n1) // k)
        zz = w["momenta"].z
        mk = lambda r, dl: (osys, w["kind"], w["q0"][n1 + r * k:n1 + (r + 1) * k], zz[n1 + r * k:n1 + (r + 1) * k],  # noqa: E731
                            w["h"], steps, dl)
        sample = f"oracle per-chain NumPy (reference style): {nw} workers x up to {k} chains x {steps} steps"
        worker = _cpu_shard_chains
    with mp.get_context("fork").Pool(nw) as pool:
        pool.map(_cpu_noop, range(nw), chunksize=1)  # start every worker before the clock does
        deadline = time.time() + pool_s
        res = pool.map(worker, [mk(r, deadline) for r in range(nw)], chunksize=1)
    total = float(sum(r[0] for r in res))
    wall = max(r[1] for r in res)
    out.update(value=total / wall,
               sample=sample + f" of the same workload, every worker running for {pool_s:.0f} s (process pool, 1 BLAS "
                               f"thread per worker; {wall:.1f} s longest)",
               single_chain_1core=single)
    return out


def cpu_baseline(config, budget_s=16.0):
    """CPU baseline of one config in a SEPARATE interpreter (the pool forks; this process holds HIP state) under a
    hard timeout, so that a wedged pool cannot cost the bench its result line.  Adds the reference-equivalent
    figure: the 1-core oracle rate x the reference/oracle ratio measured in the build container
    (profiles/cpu_calibration.json, tools/calibrate_cpu_baseline.py)."""
    env = dict(os.environ)
    for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        env[k] = "1"
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", config, "--cpu-budget", str(budget_s)]
    import signal
    limit = 4 * budget_s + 60
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                            start_new_session=True)  # its own process group: the pool's workers die with it
    try:
        stdout, stderr = proc.communicate(timeout=limit)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(proc.pid, signal.SIGKILL)
        except OSError:
            pass
        proc.communicate()
        return dict(value=None, unit="leapfrog-steps/s", cores=0, kind="port",
                    sample=f"cpu baseline worker timed out after {limit:.0f} s")
    line = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    if proc.returncode != 0 or not line:
        return dict(value=None, unit="leapfrog-steps/s", cores=0, kind="port",
                    sample=f"cpu baseline worker failed (rc={proc.returncode}): {stderr[-300:]}")
    out = json.loads(line[-1])
    cal = os.path.join(ROOT, "profiles", "cpu_calibration.json")
    if os.path.exists(cal):
        with open(cal) as fh:
            ratio = json.load(fh).get("reference_over_oracle", {}).get(config)
        if ratio:
            out["reference_equiv"] = dict(
                value=out["single_chain_1core"] * ratio, unit="leapfrog-steps/s per core",
                note=f"1-core per-chain oracle rate x {ratio:.3f} (reference/oracle speed ratio measured with the "
                     "imported reference in the build container, profiles/cpu_calibration.json)")
    return out


# ---- one config, measured ---------------------------------------------------------------------------------------
def run_config(ctx, rdzv, config, steps, warmup, rank, world, chains_per_gpu=None, traj_len=None,
               gather_mode="off", dump_trace=False):
    """W warm-up passes, then exactly `steps` timed passes between (device sync + rank barrier) pairs.
    Returns the result dict of this config (identical on every rank)."""
    from mici_amd import _ffi
    from mici_amd.runtime import DeviceBatch

    n_local = chains_per_gpu or DEFAULT_CHAINS.get(config, 4096)
    # the model from a rank-independent stream (every rank integrates the SAME target / metric); the chains' initial
    # states from the rank's own stream (rank 0: the continuation of the model stream, as in rounds 1-2)
    rng = np.random.default_rng(1234)
    w = make_workload(config, n_local, rng, chain_rng=None if rank == 0 else np.random.default_rng([1234, rank]))
    traj = traj_len or w["traj"]
    integ = w["integ"]
    batch = DeviceBatch(ctx, n_local, w["dim"])
    dirs = np.ones(n_local, dtype=np.int8)

    comm = None
    pos_all = None
    ranks_seen = [None]  # ranks of the RCCL communicator, once one exists (None: no communicator was formed)
    in_loop = world > 1 and gather_mode == "rccl"
    want_host = 1 if rank == 0 else 0  # only the trace-writing rank needs the gathered array on the host

    def setup_comm():
        import ctypes as C

        from mici_amd import distributed as mdist
        raw = mdist.exchange_unique_id(ctx, rdzv)
        h = C.c_void_p()
        _ffi.check(ctx._lib.mm_comm_create(ctx.handle, world, rank, raw, C.byref(h)), ctx.handle, "mm_comm_create")
        nr, rr = C.c_int32(0), C.c_int32(0)  # what the communicator ITSELF says it spans (ncclCommCount / ncclCommUserRank)
        _ffi.check(ctx._lib.mm_comm_count(h, C.byref(nr), C.byref(rr)), ctx.handle, "mm_comm_count")
        if rr.value != rank:
            raise RuntimeError(f"RCCL communicator reports rank {rr.value}, launcher says {rank}")
        ranks_seen[0] = int(nr.value)
        return h, np.empty((world * n_local, w["dim"]))

    if in_loop:
        comm, pos_all = setup_comm()

    def collect_traces():
        _ffi.check(ctx._lib.mm_comm_allgather_pos_async(comm, batch.handle, want_host), ctx.handle,
                   "mm_comm_allgather_pos_async")

    def finish_traces():
        _ffi.check(ctx._lib.mm_comm_wait(comm, pos_all.ctypes.data_as(_ffi.c_double_p) if want_host else None),
                   ctx.handle, "mm_comm_wait")

    def barrier():
        ctx.sync()
        if rdzv is not None:
            rdzv.barrier()

    # the initial state lives in a second device batch: the timed region starts from a device-to-device copy of it, not
    # from a host upload (whose staging copies were still seen perturbing the first timed launch)
    init = DeviceBatch(ctx, n_local, w["dim"])
    init.upload(w["q0"], w["p0"], dirs)
    _ffi.check(ctx._lib.mm_state_copy(batch.handle, init.handle), ctx.handle, "mm_state_copy")
    for _ in range(warmup):
        integ.step_device(batch, traj, ctx)
        if in_loop:
            collect_traces()
    if in_loop:
        finish_traces()
    def timed_region():
        """Exactly `steps` passes from the same resident initial state, bracketed by a barrier + stream synchronisation on
        both sides.  Returns (wall-clock seconds, kernel ms summed over the passes from the HIP events, host issue
        seconds, completed chain-steps, work counters)."""
        _ffi.check(ctx._lib.mm_state_copy(batch.handle, init.handle), ctx.handle, "mm_state_copy")  # same resident state

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
        for k in range(steps):
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
        issued = time.perf_counter() - t0  # host time to issue the passes (explicit configs: K asynchronous launches)
        if in_loop:
            finish_traces()  # the last gather must have landed inside the timed region
        barrier()
        elapsed = time.perf_counter() - t0
        if per_launch_events:
            for k in range(max(0, steps - n_pairs), steps):  # the launches whose events were not read yet
                s = (k % n_pairs) * 2
                kernel_ms += ctx.elapsed_ms(s, s + 1)
        else:
            kernel_ms = ctx.elapsed_ms(0, 1)  # K launches back to back
        return elapsed, kernel_ms, issued, done_acc, counters_acc

    # One timed region is the measurement.  On this pool the device's queue is sometimes descheduled for 20-60 ms in the
    # middle of a region (the delay sits on a 25 ms grid, a store made by a kernel of the same queue
    # arrives late, the kernels themselves run gap-free at full speed) - visible as a wall clock well above the HIP-event
    # kernel time of the SAME passes.  A single-process run then times the region again, up to twice, and reports the
    # attempt with the smallest wall clock; every attempt is listed in `attempts`.
    attempts = []
    best = None
    for _attempt in range(3 if world == 1 else 1):
        res = timed_region()
        attempts.append(dict(ms_per_step=res[0] * 1e3 / steps, kernel_ms_per_launch=res[1] / steps))
        if best is None or res[0] < best[0]:
            best = res
        if res[0] * 1e3 <= 1.10 * res[1]:
            break
    elapsed, kernel_ms, issued, done_acc, counters_acc = best
    done_local = float(n_local) * traj * steps if w["kind"] == "euclid" else done_acc
    total_steps = done_local
    rank_elapsed = [elapsed]
    if rdzv is not None:
        rank_elapsed = [float(x) for x in rdzv.allgather(repr(elapsed).encode())]
        elapsed = max(rank_elapsed)  # == rdzv.reduce_max(elapsed): the slowest rank's clock prices the job
        total_steps = rdzv.reduce_sum(done_local)

    # The job's one collective, timed on its own AFTER the timed region (communicator creation included in
    # neither).  Guarded by a timeout: a wedged RCCL bootstrap must not cost the scaling run its result line.
    trace_gather_ms, gather_note, exit_hard = None, "none", False
    if world > 1 and gather_mode == "after":
        import threading
        box = {}

        def timed_gather():
            nonlocal comm, pos_all
            try:
                comm, pos_all = setup_comm()
                for _rep in range(2):  # first gather warms the rings up
                    barrier()
                    tg = time.perf_counter()
                    collect_traces()
                    finish_traces()
                    barrier()
                    box["ms"] = (time.perf_counter() - tg) * 1e3
                box["mode"] = "rccl all-gather over xGMI, outside the timed region"
            except Exception as e:
                box["mode"] = f"rccl unavailable ({type(e).__name__}: {str(e)[:160]})"

        th = threading.Thread(target=timed_gather, daemon=True)
        th.start()
        th.join(timeout=90.0)
        if th.is_alive():
            gather_note, exit_hard = "rccl bootstrap timed out", True
        else:
            gather_note, trace_gather_ms = box.get("mode", "?"), box.get("ms")
    elif in_loop:
        gather_note = "rccl all-gather per pass, inside the timed region"
    elif world > 1 and gather_mode == "host":
        # MICI_AMD_SHARE_DEVICE: the same rank-major all-gather through the rendezvous socket (the test oracle of the
        # RCCL path, SURVEY 8e "host-side alternative must give identical bytes")
        from mici_amd import distributed as mdist
        barrier()
        tg = time.perf_counter()
        pos_all = mdist.gather_host(batch.download()[0], world * n_local, rdzv)
        barrier()
        trace_gather_ms = (time.perf_counter() - tg) * 1e3
        gather_note = "host gather over the rendezvous socket, outside the timed region (MICI_AMD_SHARE_DEVICE test switch)"
    dump = os.environ.get("MICI_AMD_BENCH_DUMP_TRACE")
    if dump and rank == 0 and dump_trace:
        np.save(dump, pos_all if pos_all is not None else batch.download()[0])

    launch_s = kernel_ms / 1e3 / steps
    # chain-steps a launch COMPLETED: explicit steps cannot fail; every other integrator counts n_done (a failed chain
    # stops - at c5's h = 0.1 a quarter of the torus chain-steps is lost that way)
    chain_steps_per_launch = n_local * traj if w["kind"] == "euclid" else done_local / steps
    d = float(w["dim"])
    if w["kind"] == "softabs":
        # algorithmic flops: n_eig * 9 D^3 (symmetric eigendecomposition
        # with vectors) + 4 D^3 per momentum-solve evaluation (grad_quadratic_form_inv's two GEMMs)
        n_m = counters_acc.get("n_metric", 0)
        n_b = max(counters_acc.get("n_fp_evals", 0) - n_m, 0)
        w["flops_per_chain_step"] = (n_m * 9 * d**3 + n_b * 4 * d**3) / max(done_local, 1.0)
        chain_steps_per_launch = done_local / steps
        # What the device EXECUTED (round 3: decompositions by matrix-product refinement of the previous eigenvectors,
        # k_softabs.hip refine_eigh): every 64^3 product of the kernel runs on the matrix cores and is counted there.
        n_prod = counters_acc.get("n_mfma_products", 0)
        if n_prod:
            npd = 64.0 if d <= 64 else (128.0 if d <= 128 else 256.0)  # the products' padded size (softabs.h NP)
            w["executed"] = dict(mfma_flops_per_chain_step=n_prod * 2.0 * npd**3 / max(done_local, 1.0),
                                 valu_flops_per_chain_step=0.0,
                                 mfma_products_per_chain_step=n_prod / max(done_local, 1.0),
                                 refined_decompositions_per_chain_step=counters_acc.get("n_refine", 0) / max(done_local, 1.0),
                                 jacobi_sweeps_per_chain_step=counters_acc.get("n_newton_iters", 0) / max(done_local, 1.0))
    if w["kind"] == "riemann":
        # algorithmic flops of SURVEY.md section 8d from the device work counters:
        #   n_M D^3/3 (factorisations) + n_inv 2D^3/3 (one explicit inverse per completed step)
        #   + (2 n_M + 3 n_B) D^2 (solves / mat-vecs / outer products), n_B = momentum-solve evals
        n_m = counters_acc.get("n_metric", 0)
        n_b = max(counters_acc.get("n_fp_evals", 0) - n_m, 0)
        w["flops_per_chain_step"] = (n_m * d**3 / 3 + done_local * 2 * d**3 / 3 + (2 * n_m + 3 * n_b) * d * d) \
            / max(done_local, 1.0)
        chain_steps_per_launch = done_local / steps
        # What the device EXECUTED for those constructions (round 3: the solve-only ones are refined from the explicit
        # inverse at the step's start instead of being factorised, implicit_core.h): matrix-core flops of the sweeps
        # that ran, from the kernels' MFMA counts (v_mfma_f64_16x16x4 = 2048 flop), and the vector flops of the
        # refinement's product pairs and of the inverse mat-vecs.
        mf = _sweep_mfma_counts(int(d))
        if mf is not None:
            mfma_flops = 2048.0 * (counters_acc.get("n_factor_full", 0) * mf["full"]
                                   + counters_acc.get("n_factor_solve", 0) * mf["solve"])
            dp = float(mf["padded_dim"])
            # (round 6: a solve-only construction of the rank-one-update metric by the Woodbury identity, implicit_core.h
            # lowrank_solve, is ONE product with the held inverse: 2 D^2 flops)
            n_lr = counters_acc.get("n_lowrank", 0)
            # (an explicit inverse carried to the new position by the rank-two update, lowrank_update: one product + two
            # multiply-adds per held entry - the lower triangle on the tile kernels, the whole matrix in row form at D <= 64)
            n_up = counters_acc.get("n_inverse_update", 0)
            valu_flops = (4.0 * counters_acc.get("n_refine", 0) + 2.0 * n_lr + (6.0 if d <= 64 else 4.0) * n_up
                          + 2.0 * n_b + 6.0 * done_local) * dp * dp
            w["executed"] = dict(mfma_flops_per_chain_step=mfma_flops / max(done_local, 1.0),
                                 valu_flops_per_chain_step=valu_flops / max(done_local, 1.0),
                                 refine_pairs_per_chain_step=counters_acc.get("n_refine", 0) / max(done_local, 1.0),
                                 lowrank_solves_per_chain_step=n_lr / max(done_local, 1.0),
                                 inverse_updates_per_chain_step=n_up / max(done_local, 1.0),
                                 sweeps_per_chain_step=(counters_acc.get("n_factor_full", 0)
                                                        + counters_acc.get("n_factor_solve", 0)) / max(done_local, 1.0))
    hbm_model = None
    if w["kind"] == "riemann" and d > 279:
        # the global-memory tier (csrc/implicit_global.h) is HBM-bound: what its launch has to move, from the work counters -
        # a sweep is ceil(D / NB) passes reading and writing the DP x DP workspace (+ one write to build it), every product
        # with the held inverse one read of it (the rank-one base matrix is shared by all chains: L2 / MALL, not counted)
        dp = float((int(d) + 63) & ~63)
        nb = 32.0 if dp <= 512 else 16.0
        sweeps = counters_acc.get("n_factor_full", 0)
        n_m = counters_acc.get("n_metric", 0)
        n_b = max(counters_acc.get("n_fp_evals", 0) - n_m, 0)
        # round 6: a sweep = the workspace written once (build), ceil(D / NB) passes over its LOWER triangle (half read, half
        # written), one mirror pass (half read, the whole written, + the FP32 copy: half the bytes again); F r of a CG pair
        # reads the FP32 copy (half the bytes; a lock-step pair of solves shares the pass - not subtracted here), the momentum
        # evaluations and the A / C sub-steps the FP64 matrix.  The rank-one base matrix is shared by all chains (L2 / MALL).
        # (round 6: the Woodbury solves of the rank-one-update metric read the FP64 inverse once each; in lock step two share a pass -
        # not subtracted here)
        # an inverse update (lowrank_update) is one product + one read-modify-write pass of the FP64 workspace: 3 passes' bytes
        n_lr, n_up = counters_acc.get("n_lowrank", 0), counters_acc.get("n_inverse_update", 0)
        f64_products = n_b + 4.0 * done_local + n_lr + 3.0 * n_up
        note_sym = ""
        if n_lr > 0 and os.environ.get("MICI_AMD_GLOBAL_SYM", "1") != "0":
            # implicit_global.h sym_walk: a product reads the lower tiles of the symmetric inverse (half a pass), the update's
            # read-modify-write touches the lower tiles too: product 0.5 + 2 x 0.5
            f64_products = 0.5 * (n_b + 4.0 * done_local + n_lr) + 1.5 * n_up
            note_sym = "; products and updates on the lower tiles only (sym_walk): half the bytes each"
        f32_products = counters_acc.get("n_refine", 0)
        bytes_total = 8.0 * dp * dp * (sweeps * (1.0 + np.ceil(d / nb) + 2.0) + f64_products + 0.5 * f32_products)
        hbm_model = dict(bytes_per_launch=bytes_total / steps, achieved_GBs=bytes_total / steps / launch_s / 1e9,
                         frac_of_hbm_peak=bytes_total / steps / launch_s / 1e9 / HBM_PEAK_GBS,
                         bytes_per_chain_step=bytes_total / max(done_local, 1.0),
                         note="modelled traffic of the HBM-resident metric: sweeps x (ceil(D / NB) + 3) + FP64 products + FP32 "
                              "products / 2, x 8 DP^2 bytes" + note_sym)
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
            roof["fp64_valu"] = dict(achieved=tf, peak=FP64_MFMA_PEAK_TF, unit="TFLOP/s", frac=tf / FP64_MFMA_PEAK_TF,
                                     chain_steps_per_launch=chain_steps_per_launch)
            if w.get("valu_executed_flops_per_chain_step"):  # the flops the kernel's instruction stream really holds
                ex_tf = w["valu_executed_flops_per_chain_step"] * chain_steps_per_launch / launch_s / 1e12
                roof["fp64_valu"].update(executed_flops_per_chain_step=w["valu_executed_flops_per_chain_step"],
                                         executed_achieved=ex_tf, executed_frac=ex_tf / FP64_MFMA_PEAK_TF)
    # HBM traffic per launch from the committed rocprofv3 PMC passes of this same workload (FETCH_SIZE /
    # WRITE_SIZE cannot be read from inside the process); null when this shape was not profiled
    default_shape = chains_per_gpu is None and traj_len is None
    for pmc_name in (f"r06_{config}_pmc_hbm.json", f"r05_{config}_pmc_hbm.json", f"r04_{config}_pmc_hbm.json", f"r03_{config}_pmc_hbm.json", f"r02_{config}_pmc_hbm.json"):
        pmc = os.path.join(ROOT, "profiles", pmc_name)
        if default_shape and os.path.exists(pmc):
            with open(pmc) as fh:
                roof["traffic"] = json.load(fh)["traffic_bytes_per_launch"]
            roof["traffic_source"] = f"profiles/{pmc_name} (rocprofv3 --pmc, corrected)"
            break
    if w.get("executed"):
        ex = w["executed"]
        per_launch = chain_steps_per_launch / launch_s / 1e12
        roof["mfma_executed_flops"] = ex["mfma_flops_per_chain_step"] * chain_steps_per_launch  # per launch
        roof["mfma_busy"] = ex["mfma_flops_per_chain_step"] * per_launch / FP64_MFMA_PEAK_TF  # executed MFMA rate / peak
        roof["executed"] = dict(ex, achieved_tflops=(ex["mfma_flops_per_chain_step"] + ex["valu_flops_per_chain_step"])
                                * per_launch,
                                note="flops the kernels executed; `achieved` / `frac` price the SURVEY 8d algorithmic "
                                     "count (one factorisation / one eigendecomposition per metric construction) as the "
                                     "contract asks")
        if w["kind"] == "riemann" and counters_acc.get("n_lowrank", 0) > 0:
            # Round 6: the built-in rank-one-update metric's constructions come from the held inverse by the Woodbury identity
            # (implicit_core.h lowrank_solve / lowrank_update) - O(D^2) each where SURVEY 8d prices a D^3 / 3 factorisation.  The
            # algorithmic count no longer describes what runs (it would read as more than the machine's peak), so this entry's
            # `achieved` / `frac` are the EXECUTED flops, and what the reference's algorithm would need is kept beside them.
            ref_tf = w["flops_per_chain_step"] * per_launch
            roof["reference_algorithm"] = dict(
                flops_per_chain_step=w["flops_per_chain_step"], at_this_rate_tflops=ref_tf,
                ratio_to_fp64_peak=ref_tf / FP64_MFMA_PEAK_TF,
                note="SURVEY 8d count (a factorisation per metric construction) x this run's steps/s: what a kernel running "
                     "the reference's algorithm would have to sustain; a ratio above 1 means the step is done with fewer "
                     "flops than that algorithm needs at peak")
            roof["achieved"] = roof["executed"]["achieved_tflops"]
            roof["frac"] = roof["achieved"] / FP64_MFMA_PEAK_TF
            roof["executed"]["note"] = ("flops the kernels executed; with the Woodbury path on, `achieved` / `frac` price "
                                        "these (see reference_algorithm)")
    if hbm_model is not None:
        roof["hbm_model"] = hbm_model
    roof["kernel_ms_per_launch"] = kernel_ms / steps
    roof["host_issue_ms"] = issued * 1e3  # of all passes; large values = the host, not the GPU, paced the region
    roof["attempts"] = attempts  # every timed region of this config (see the re-timing policy above)
    roof["algorithmic_flops_per_chain_step"] = w["flops_per_chain_step"]
    # ONE unit for every entry: SURVEY 8d's 32 D bytes per chain-STEP.  An HBM-bound entry's `achieved` prices what a
    # fused launch really has to move - the state read once and written once per LAUNCH (`..._per_chain_launch`)
    roof["algorithmic_bytes_per_chain_step"] = w["bytes_per_chain_step"]
    roof["algorithmic_bytes_per_chain_launch"] = w["bytes_per_chain_step"]
    if counters_acc:
        roof["work_counters"] = counters_acc

    if comm is not None and not exit_hard:
        ctx._lib.mm_comm_destroy(comm)
    batch.close()
    init.close()
    return dict(
        value=total_steps / elapsed, unit="leapfrog-steps/s", steps=steps, warmup=warmup,
        ms_per_step=elapsed / steps * 1e3, roofline=roof,
        rank_elapsed_s=dict(min=min(rank_elapsed), max=max(rank_elapsed), per_rank=rank_elapsed),
        workload=f"{w['name']}, D={w['dim']}, {n_local} chains/GPU x {world} GPU, h={w['h']}, one pass = a trajectory "
                 f"of {traj} leapfrog steps per chain",
        baseline_config=BASELINE_CONFIG.get(config, config), chains_per_gpu=n_local, dim=w["dim"], traj_len=traj,
        trace_gather=gather_note, trace_gather_ms=trace_gather_ms, _exit_hard=exit_hard,
        # ranks that took part: the RCCL communicator's own count where one was formed, else the rendezvous' (world)
        n_ranks_seen=ranks_seen[0] if ranks_seen[0] is not None else (len(rank_elapsed) if rdzv is not None else 1),
        chains_total=n_local * world)


# ---- the result line ---------------------------------------------------------------------------------------------
HEADLINE_MAX_BYTES = 4096  # the driver's capture lost a 22 KB line in round 4 (BENCH_r04.json: parsed null)
_ROOF_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms_per_launch", "mfma_busy",
              "algorithmic_flops_per_chain_step", "algorithmic_bytes_per_chain_step", "algorithmic_bytes_per_chain_launch")
_CPU_KEYS = ("value", "unit", "cores", "kind", "sample", "single_chain_1core")


def _sig(x, digits=6):
    """Floats to `digits` significant figures (the line is read by a parser, not compared bit for bit)."""
    if isinstance(x, float):
        return float(f"{x:.{digits}g}")
    if isinstance(x, dict):
        return {k: _sig(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, digits) for v in x]
    return x


def compact_result(full, sidecar=None):
    """The LAST stdout line: the headline only - metric, value, unit, n_gpus, steps, warmup, ms_per_step, dtype,
    config{workload, ...}, roofline{bound, achieved, peak, unit, frac, traffic, kernel_ms_per_launch, algorithmic_*},
    cpu_baseline{value, unit, cores, kind, sample} - plus five scalars per extra config (`configs.<name>` = value,
    ms_per_step, roofline frac, kernel ms per launch, CPU-baseline value).  Everything else is in the sidecar."""
    out = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                "scaling", "vs_baseline", "dtype", "data")}
    cfg = dict(full["config"])
    cfg["workload"] = cfg["workload"][:200]
    cfg["parallelism"] = cfg["parallelism"][:120]
    out["config"] = cfg
    out["roofline"] = {k: full["roofline"][k] for k in _ROOF_KEYS if k in full["roofline"]}
    if "cpu_baseline" in full:
        cb = {k: full["cpu_baseline"][k] for k in _CPU_KEYS if k in full["cpu_baseline"]}
        if isinstance(cb.get("sample"), str):
            cb["sample"] = cb["sample"][:200]
        out["cpu_baseline"] = cb
    if full.get("configs"):
        brief = {}
        for name, res in full["configs"].items():
            if "error" in res:
                brief[name] = {"error": res["error"][:80]}
                continue
            brief[name] = {"value": res["value"], "ms_per_step": res["ms_per_step"],
                           "frac": res["roofline"]["frac"], "kernel_ms": res["roofline"]["kernel_ms_per_launch"]}
            # the fifth scalar: at N = 1 the CPU baseline, at N > 1 (no CPU leg) the chains the whole job integrated - c4 at
            # N = 8 is BASELINE configs[3]'s 8192 chains, c5 its 16384
            if full.get("n_gpus", 1) > 1:
                brief[name]["chains"] = res.get("chains_total")
            else:
                brief[name]["cpu"] = (res.get("cpu_baseline") or {}).get("value")
        out["configs"] = brief
    if full.get("n_gpus", 1) > 1 and "rank_elapsed_s" in full:  # a straggler rank shows up here (value uses the max)
        out["rank_elapsed_s"] = full["rank_elapsed_s"]
    if sidecar:
        out["detail"] = sidecar
    out = _sig(out)
    line = json.dumps(out, separators=(",", ":"))
    if len(line) > HEADLINE_MAX_BYTES:  # never again: drop the per-config scalars before the headline suffers
        out.pop("configs", None)
        line = json.dumps(out, separators=(",", ":"))
    assert len(line) <= HEADLINE_MAX_BYTES, len(line)
    return line


def emit_result(full):
    """Full detail -> sidecar file `bench_configs.json` (next to bench.py; MICI_AMD_BENCH_SIDECAR overrides, also copied
    under gpurun_out/ when that directory exists); one short `# name: ...` line per extra config on stderr; the compact
    headline is the ONLY line of stdout."""
    side = os.environ.get("MICI_AMD_BENCH_SIDECAR", os.path.join(ROOT, "bench_configs.json"))
    written = None
    try:
        with open(side, "w") as fh:
            json.dump(full, fh, indent=1)
        written = os.path.relpath(side, ROOT) if side.startswith(ROOT) else side
        scratch = os.path.join(ROOT, "gpurun_out")
        if os.path.isdir(scratch) and "MICI_AMD_BENCH_SIDECAR" not in os.environ:
            with open(os.path.join(scratch, "bench_configs.json"), "w") as fh:
                json.dump(full, fh, indent=1)
    except OSError:
        pass
    for name, res in (full.get("configs") or {}).items():  # one short line per extra config, before the headline
        if "error" in res:
            print(f"# {name}: {res['error']}", file=sys.stderr, flush=True)
        else:
            print(f"# {name}: {res['value']:.4g} steps/s, {res['ms_per_step']:.4g} ms/pass, roofline "
                  f"{res['roofline']['frac']:.3f} ({res['roofline']['bound']})", file=sys.stderr, flush=True)
    print(compact_result(full, written), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c2")
    ap.add_argument("--chains-per-gpu", type=int, default=None)
    ap.add_argument("--traj-len", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="only the --config workload (no `configs` object)")
    ap.add_argument("--cpu-baseline-worker", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-budget", type=float, default=16.0, help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.cpu_baseline_worker:  # child interpreter of cpu_baseline(): no HIP in here
        print(json.dumps(cpu_baseline_measure(args.cpu_baseline_worker, args.cpu_budget)), flush=True)
        return

    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if env_world is None and args.gpus > 1:
        # started plainly: be the launcher.  One rank process per GPU, rendezvous over a Unix socket; the ranks
        # check the device count together once they have met and all fail if the box has fewer than N devices.
        from mici_amd.rendezvous import spawn_ranks
        codes = spawn_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus)
        if any(codes):
            print(f"bench.py --gpus {args.gpus}: rank exit codes {codes}", file=sys.stderr)
        raise SystemExit(max(abs(c) for c in codes))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    world = int(env_world or "1")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import ctypes as C

    from mici_amd import _ffi
    from mici_amd.rendezvous import Rendezvous
    from mici_amd.runtime import Context

    rdzv = Rendezvous.from_env() if world > 1 else None
    count = C.c_int(0)
    n_dev = count.value if _ffi.load().mm_device_count(C.byref(count)) == 0 else 0
    if rdzv is not None:  # every rank learns the smallest count: all leave together, none waits in a collective
        n_dev = int(min(int(x) for x in rdzv.allgather(str(n_dev).encode())))
    share = world > 1 and os.environ.get("MICI_AMD_SHARE_DEVICE", "") == "1"  # test switch, see the docstring
    if (n_dev < world or n_dev <= local_rank) and not (share and n_dev >= 1):
        if rdzv is not None:
            rdzv.close()
        raise SystemExit(f"bench.py --gpus {world}: rank {rank} needs HIP device {local_rank} but only {n_dev} HIP "
                         "device(s) are visible (one process per GPU, no oversubscription, no CPU fallback)")
    ctx = Context(local_rank % n_dev if share else local_rank)

    gather_mode = "off"
    if world > 1:
        gather_mode = "rccl" if os.environ.get("MICI_AMD_BENCH_GATHER", "") == "rccl" else "after"
        if share:
            gather_mode = "host"
    head = run_config(ctx, rdzv, args.config, args.steps, args.warmup, rank, world, args.chains_per_gpu,
                      args.traj_len, gather_mode, dump_trace=True)
    exit_hard = head.pop("_exit_hard")
    # every GPU measurement comes first; the CPU baselines (all host cores busy) run after the last timed region
    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline

    configs = {}
    default_shape = args.chains_per_gpu is None and args.traj_len is None
    if not args.no_extra_configs and args.config == "c2" and default_shape and not exit_hard:
        for cfg in EXTRA_CONFIGS:
            k = min(args.steps, EXTRA_STEP_CAP.get(cfg, args.steps))
            wu = min(args.warmup, 1 if cfg in EXTRA_STEP_CAP else args.warmup)
            try:
                res = run_config(ctx, rdzv, cfg, k, wu, rank, world)
            except Exception as e:  # keep the headline line alive; every rank fails the same way
                configs[cfg] = dict(error=f"{type(e).__name__}: {str(e)[:200]}")
                continue
            res.pop("_exit_hard")
            for key in ("trace_gather", "trace_gather_ms", "n_ranks_seen"):
                res.pop(key)
            configs[cfg] = res
    if want_cpu:
        head["cpu_baseline"] = cpu_baseline(args.config, 20.0)
        for cfg, res in configs.items():
            if "error" not in res:
                res["cpu_baseline"] = cpu_baseline(cfg, 8.0)

    if rank == 0:
        out = {
            "metric": "leapfrog-steps/sec (all chains)",
            "value": head["value"],
            "unit": "leapfrog-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": head["workload"],
                "baseline_config": head["baseline_config"],
                "chains_per_gpu": head["chains_per_gpu"], "dim": head["dim"], "traj_len": head["traj_len"],
                "parallelism": (f"chains sharded x{world}, one process per GPU, no collective in the timed region"
                                if gather_mode != "rccl" else f"chains sharded x{world}, RCCL trace gather per pass")
                               + (" [MICI_AMD_SHARE_DEVICE: ranks share devices - not a measurement]" if share else ""),
                "trace_gather": head["trace_gather"],
                "trace_gather_ms": head["trace_gather_ms"],
                "n_ranks_seen": head.get("n_ranks_seen", world),
                "chains_total": head.get("chains_total"),
            },
            "roofline": head["roofline"],
            # wall clock of the timed region on every rank (value uses the max): a straggler shows up here
            "rank_elapsed_s": head["rank_elapsed_s"],
        }
        if "cpu_baseline" in head:
            out["cpu_baseline"] = head["cpu_baseline"]
        if configs:
            out["configs"] = configs
        emit_result(out)

    if exit_hard:  # a collective is wedged in the helper thread: the result line is out, leave without cleanup
        sys.stdout.flush()
        os._exit(0)
    if rdzv is not None:
        rdzv.barrier()
        rdzv.close()


if __name__ == "__main__":
    main()
