#!/usr/bin/env python3
"""Constrained leapfrog on 8 < D <= 64 systems: the wave-per-chain kernel (k_constrained_wave.hip) against the
lane-per-chain core with its per-chain arrays in scratch (MICI_AMD_CONSTRAINED_KERNEL=lane), same inputs.
    python tools/ubench_constrained_wide.py            # runs itself twice (the switch is read once per process)"""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CASES = [(16, 2, "dense", 4096), (32, 4, "dense", 4096), (64, 8, "diag", 4096), (64, 8, "dense", 4096), (64, 8, "dense", 64)]


def worker():
    from mici_amd import integrators, models, systems
    from mici_amd.runtime import DeviceBatch, default_context
    from oracle import models as omdl
    out = {}
    ctx = default_context()
    for d, c, mk, n in CASES:
        rng = np.random.default_rng(d * 100 + c)
        a, b = rng.standard_normal((c, d)), rng.standard_normal(c)
        metric = np.exp(0.2 * rng.standard_normal(d)) if mk == "diag" else omdl.make_spd(d, rng)
        system = systems.DenseConstrainedEuclideanMetricSystem(models.Poly(d, 1.0, 0.25), models.LinearConstr(a, b), metric=metric)
        integ = integrators.ConstrainedLeapfrogIntegrator(system, 0.05)
        x0 = np.linalg.lstsq(a, b, rcond=None)[0]
        z = rng.standard_normal((n, d))
        null = np.eye(d) - a.T @ np.linalg.solve(a @ a.T, a)
        q0 = x0 + 0.5 * z @ null.T
        p0 = system.sample_momentum_batch(q0, rng.standard_normal((n, d)))
        batch = DeviceBatch(ctx, n, d)
        steps = 20
        best = 1e9
        for rep in range(3):
            batch.upload(q0, p0, np.ones(n, dtype=np.int8))
            ctx.sync()
            t0 = time.perf_counter()
            integ.step_device(batch, steps, ctx)
            ctx.sync()
            best = min(best, time.perf_counter() - t0)
        q, p, _ = batch.download()
        st, nd = batch.download_status()
        out[f"D={d} C={c} {mk} N={n}"] = dict(ms=best * 1e3, steps_per_s=n * steps / best, ok=float((st == 0).mean()),
                                              checksum=float(np.abs(q).sum() + np.abs(p).sum()))
        batch.close()
    print(json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "worker":
        worker()
        sys.exit(0)
    res = {}
    for mode in ("wave", "lane"):
        env = dict(os.environ)
        if mode == "lane":
            env["MICI_AMD_CONSTRAINED_KERNEL"] = "lane"
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "worker"], capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        res[mode] = json.loads(r.stdout.strip().splitlines()[-1])
    print(f"{'case':28s} {'wave per chain':>22s} {'lane per chain (scratch)':>26s}   ratio   |rel. checksum difference|")
    for k in res["wave"]:
        a, b = res["wave"][k], res["lane"][k]
        print(f"{k:28s} {a['steps_per_s']:12.3e} steps/s {b['steps_per_s']:16.3e} steps/s   {a['steps_per_s'] / b['steps_per_s']:6.1f}x"
              f"   {abs(a['checksum'] - b['checksum']) / b['checksum']:.1e}  (ok {a['ok']:.2f} / {b['ok']:.2f})")
