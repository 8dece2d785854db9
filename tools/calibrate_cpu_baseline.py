#!/usr/bin/env python3
"""Calibrate the CPU baseline (build container only): speed of the IMPORTED reference relative to the oracle.

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 OMP_NUM_THREADS=1 OPENBLAS_NUM_THREADS=1 \
        python /root/repo/tools/calibrate_cpu_baseline.py

For every bench config the same synthetic workload (bench.make_workload) is stepped one chain
at a time on one core by (1) the reference's own System + Integrator classes and (2) the oracle's per-chain
functions; the ratio reference-steps/s : oracle-steps/s goes to profiles/cpu_calibration.json.  On the GPU box -
where the reference cannot travel - bench.py multiplies the oracle's 1-core per-chain rate measured there by this
ratio and reports it as `cpu_baseline.reference_equiv`.
"""

from __future__ import annotations

import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/src")

import mici  # noqa: E402  (the reference)
from mici.states import ChainState  # noqa: E402

import bench  # noqa: E402
from oracle import integrators as orc  # noqa: E402


def reference_pair(config, w, osys):
    """The reference's system + integrator for a bench config, built from the oracle's closed-form model."""
    t = osys.target
    if w["kind"] == "euclid":
        metric = None if osys.metric is None else np.array(osys.metric)
        system = mici.systems.EuclideanMetricSystem(neg_log_dens=t.neg_log_dens, grad_neg_log_dens=t.grad, metric=metric)
        if config == "c2bcss":
            return system, mici.integrators.BCSSThreeStageIntegrator(system, w["h"])
        return system, mici.integrators.LeapfrogIntegrator(system, w["h"])
    if w["kind"] == "riemann":
        system = mici.systems.DenseRiemannianMetricSystem(
            neg_log_dens=t.neg_log_dens, grad_neg_log_dens=t.grad, metric_func=osys.rmetric.metric_func,
            vjp_metric_func=osys.rmetric.vjp_metric_func)
        return system, mici.integrators.ImplicitLeapfrogIntegrator(system, w["h"])
    if w["kind"] == "softabs":
        system = mici.systems.SoftAbsRiemannianMetricSystem(
            neg_log_dens=t.neg_log_dens, grad_neg_log_dens=t.grad, hess_neg_log_dens=t.hess, mtp_neg_log_dens=t.mtp,
            softabs_coeff=osys.softabs_coeff)
        return system, mici.integrators.ImplicitLeapfrogIntegrator(system, w["h"])
    c = osys.constraint
    system = mici.systems.DenseConstrainedEuclideanMetricSystem(
        neg_log_dens=t.neg_log_dens, grad_neg_log_dens=t.grad, constr=c.constr, jacob_constr=c.jacob_constr)
    return system, mici.integrators.ConstrainedLeapfrogIntegrator(system, w["h"])


def main():
    budget = float(os.environ.get("CALIBRATE_SECONDS", "6"))
    out = {"reference_over_oracle": {}, "detail": {}, "note": (
        "1 core, one chain at a time, OMP/OPENBLAS threads = 1, build container; reference = /root/reference/src "
        "imported; oracle = oracle.integrators per-chain functions; same inputs")}
    for config in ("c2", "c2i", "c2iv", "c2bcss", "c3", "c3b", "c4", "c5"):
        n = 64
        w = bench.make_workload(config, n, np.random.default_rng(1234), device=False)
        osys = w["make_oracle"]()
        if w.get("momenta") is not None:
            w["momenta"].fix(0, n)
        system, integ = reference_pair(config, w, osys)
        steps = {"euclid": 200, "riemann": 2 if w["dim"] > 128 else 5, "softabs": 3, "constrained": 50}[w["kind"]]
        # reference
        t0, done, c = time.perf_counter(), 0, 0
        while time.perf_counter() - t0 < budget and c < n:
            state = ChainState(pos=w["q0"][c].copy(), mom=w["p0"][c].copy(), dir=1)
            for _ in range(steps):
                state = integ.step(state)
            done += steps
            c += 1
        ref_rate = done / (time.perf_counter() - t0)
        # oracle, same chains
        coefs = w.get("coefficients")
        if w["kind"] == "euclid":
            if coefs is not None:
                free = list(coefs)[:(len(coefs) - 3) // 2]
                fn = lambda q, p: orc.composition_steps(osys, q, p, w["h"], steps, free)  # noqa: E731
            else:
                fn = lambda q, p: orc.leapfrog_steps(osys, q, p, w["h"], steps)  # noqa: E731
        elif w["kind"] == "constrained":
            fn = lambda q, p: orc.constrained_leapfrog_steps(osys, q, p, w["h"], steps)  # noqa: E731
        else:
            fn = lambda q, p: orc.implicit_leapfrog_steps(osys, q, p, w["h"], steps)  # noqa: E731
        t0, done, c2 = time.perf_counter(), 0, 0
        while time.perf_counter() - t0 < budget and c2 < n:
            fn(w["q0"][c2], w["p0"][c2])
            done += steps
            c2 += 1
        orc_rate = done / (time.perf_counter() - t0)
        out["reference_over_oracle"][config] = ref_rate / orc_rate
        out["detail"][config] = dict(reference_steps_per_s_per_core=ref_rate, oracle_steps_per_s_per_core=orc_rate,
                                     chains=[c, c2], steps_per_chain=steps)
        print(f"{config}: reference {ref_rate:.4g} steps/s/core, oracle {orc_rate:.4g}, ratio {ref_rate / orc_rate:.3f}",
              flush=True)
    path = os.path.join(ROOT, "profiles", "cpu_calibration.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
