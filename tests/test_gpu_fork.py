"""The forked matrix-core wave kernel (csrc/implicit_fork.h, round 6) against the one-wave kernel it forks
(csrc/implicit_mfma.h): the same arithmetic on two waves must give the SAME BITS - positions, momenta, statuses, completed
steps and the reference's fixed-point counters - for chains that run through, chains whose reversibility check fails, chains
that diverge, a chain count that does not fill a workgroup, per-chain directions, and with the refinement switched off (no
fork ever happens then).  Reference arithmetic: integrators.py:493-544 (the two position solves: :521-536)."""

import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

PROG = (
    "import numpy as np, sys; sys.path.insert(0, %r)\n"
    "from mici_amd import integrators, models, systems\n"
    "out = []\n"
    "for (d, n, h, steps, kind) in ((64, 37, 0.02, 6, 'rank1'), (40, 9, 0.05, 4, 'rank1'), (64, 16, 0.35, 3, 'rank1'), (48, 8, 0.05, 3, 'diagquad')):\n"
    "    rng = np.random.default_rng(100 + d + n)\n"
    "    a = rng.standard_normal((d, d)); B = a @ a.T / d + np.eye(d)\n"
    "    metric = models.Rank1Metric(B) if kind == 'rank1' else models.DiagQuadMetric(d)\n"
    "    s = systems.DenseRiemannianMetricSystem(models.Banana(d), metric)\n"
    "    q0 = rng.standard_normal((n, d)); p0 = s.sample_momentum_batch(q0, rng.standard_normal((n, d)))\n"
    "    dirs = np.ones(n, dtype=np.int8); dirs[1::3] = -1\n"
    "    i = integrators.ImplicitLeapfrogIntegrator(s, h)\n"
    "    q, p, st, nd = i.step_batch(q0, p0, dirs, n_steps=steps)\n"
    "    c = i.last_counters\n"
    "    out += [q.ravel(), p.ravel(), st.astype(float), nd.astype(float),\n"
    "            np.array([c['n_fp_evals'], c['n_fp_solves'], c['n_metric'], c['n_grad']], dtype=float)]\n"
    "np.save(sys.argv[1], np.concatenate(out))\n" % ROOT)


def _run(tmp_path, name, **env):
    path = str(tmp_path / f"{name}.npy")
    e = dict(os.environ)
    for k in ("MICI_AMD_FORK", "MICI_AMD_DUAL", "MICI_AMD_REFINE", "MICI_AMD_PAIR"):
        e.pop(k, None)
    e["MICI_AMD_LOWRANK"] = "0"  # (the forked kernel runs the CG refinement: the rank-one metric's default is the Woodbury path)
    e.update(env)
    r = subprocess.run([sys.executable, "-c", PROG, path], env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return np.load(path)


def test_forked_kernel_gives_the_bits_of_the_one_wave_kernel(tmp_path):
    fork = _run(tmp_path, "fork")                          # the default
    one = _run(tmp_path, "one", MICI_AMD_FORK="0")         # the one-wave kernel of rounds 3-5
    idle = _run(tmp_path, "idle", MICI_AMD_DUAL="0")       # the forked kernel's code, the second wave idle
    assert np.array_equal(fork, one, equal_nan=True), "forked kernel differs from the one-wave kernel"
    assert np.array_equal(idle, one, equal_nan=True), "forked kernel with an idle second wave differs"
    # the workload is not trivial: some chains stop early (h = 0.35 on the banana), most run through
    assert np.isfinite(fork).all()


def test_forked_kernel_without_refinement_never_forks_and_agrees(tmp_path):
    a = _run(tmp_path, "fork_norefine", MICI_AMD_REFINE="0")
    b = _run(tmp_path, "one_norefine", MICI_AMD_REFINE="0", MICI_AMD_FORK="0")
    assert np.array_equal(a, b, equal_nan=True)
