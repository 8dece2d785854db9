#!/usr/bin/env python3
"""Actual device-vs-reference error of every implicit-midpoint fixture (GPU box): what tolerance the fixtures
support, per fixture.  Prints one JSON object."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import golden_names, load_golden  # noqa: E402

from mici_amd import integrators, models, solvers, systems  # noqa: E402

out = {}
for name in golden_names("midpoint"):
    g = load_golden(name)
    n, d = g["q0"].shape
    target = models.target_from_id(g["target"], g["target_params"], d)
    if str(g["system"]) == "euclid":
        mk = int(g["metric_kind"])
        system = systems.EuclideanMetricSystem(target, metric=None if mk == models.METRIC_IDENTITY else g["metric"])
    elif str(g["system"]) == "softabs":
        system = systems.SoftAbsRiemannianMetricSystem(target, softabs_coeff=float(g["rmetric_params"][0]))
    else:
        system = systems.DenseRiemannianMetricSystem(target, models.rmetric_from_id(g["rmetric"], g["rmetric_params"], d))
    norm = {0: solvers.maximum_norm, 1: solvers.euclidean_norm}[int(g["norm"])]
    fps = {0: solvers.solve_fixed_point_direct, 1: solvers.solve_fixed_point_steffensen}[int(g["fp_solver"])]
    integ = integrators.ImplicitMidpointIntegrator(
        system, float(g["step_size"]), reverse_check_norm=norm, fixed_point_solver=fps,
        fixed_point_solver_kwargs=dict(norm=norm, convergence_tol=float(g["fp_conv_tol"]),
                                       divergence_tol=float(g["fp_div_tol"]), max_iters=int(g["fp_max_iters"])))
    worst = 0.0
    for k, s in enumerate(int(s) for s in g["checkpoints"]):
        q, p, status, n_done = integ.step_batch(g["q0"], g["p0"], g["dir"], n_steps=s)
        for a, b in ((q, g["q_out"][k]), (p, g["p_out"][k])):
            a, b = np.nan_to_num(a), np.nan_to_num(b)
            worst = max(worst, float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b)))))
    out[name] = dict(max_scaled_err=worst, steps=int(g["checkpoints"].max()), fp_evals=int(g.get("count_fp_iters", -1))
                     if "count_fp_iters" in g else None)
print(json.dumps(out, indent=1))
