"""The oracle over MANY chains on the host cores of the GPU box: a process pool of fresh interpreters
(spawn - the parent holds HIP state and must not be forked) whose workers build the bench workload's oracle twin once
(bench.make_workload(..., device=False): NumPy / SciPy only) and integrate the chains they are handed, one at a time, as
the reference itself would.  Test infrastructure: nothing under mici_amd/ imports this."""

import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_W = {}  # per worker process: config -> (oracle system, kind)


def _init():
    for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[k] = "1"
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)


def _system(config, n):
    if config not in _W:
        import bench
        w = bench.make_workload(config, n, np.random.default_rng(1234), device=False)  # the model stream of rank 0
        _W[config] = (w["make_oracle"](), w["kind"])
    return _W[config]


def _chunk(job):
    config, n, q0, p0, dirs, h, steps = job
    from oracle import integrators as orc
    osys, kind = _system(config, n)
    q, p = np.empty_like(q0), np.empty_like(p0)
    status, n_done = np.zeros(len(q0), dtype=np.int64), np.zeros(len(q0), dtype=np.int64)
    for c in range(len(q0)):
        dt = float(dirs[c]) * h
        if kind == "euclid":
            q[c], p[c] = orc.leapfrog_steps(osys, q0[c], p0[c], dt, steps)
            n_done[c] = steps
        elif kind == "constrained":
            q[c], p[c], status[c], n_done[c] = orc.constrained_leapfrog_steps(osys, q0[c], p0[c], dt, steps)
        else:
            q[c], p[c], status[c], n_done[c] = orc.implicit_leapfrog_steps(osys, q0[c], p0[c], dt, steps)
    return q, p, status, n_done


def host_workers():
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:  # a container can show 256 CPUs and be throttled to a handful (bench._host_cores)
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()[:2]
        if quota != "max":
            cores = max(1, min(cores, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return max(1, min(cores, 64))


class OraclePool:
    """with OraclePool() as pool: q, p, status, n_done = pool.run(config, n_total, q0, p0, dirs, h, steps)"""

    def __init__(self, workers=None):
        self.workers = workers or host_workers()
        self.pool = mp.get_context("spawn").Pool(self.workers, initializer=_init)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.pool.terminate()
        self.pool.join()

    def run(self, config, n_total, q0, p0, dirs, h, steps, chunk=None):
        """`n_total`: the shard size the workload was made with (the model's random stream depends on nothing else, but the
        workers rebuild it the way rank 0 did).  q0 / p0: the chains to integrate (any subset of the shard)."""
        m = len(q0)
        dirs = np.broadcast_to(np.asarray(dirs, dtype=np.int8), (m,))
        chunk = chunk or max(1, min(16, -(-m // (4 * self.workers))))
        jobs = [(config, n_total, q0[i:i + chunk], p0[i:i + chunk], dirs[i:i + chunk], h, steps) for i in range(0, m, chunk)]
        out = self.pool.map(_chunk, jobs, chunksize=1)
        return (np.concatenate([o[0] for o in out]), np.concatenate([o[1] for o in out]),
                np.concatenate([o[2] for o in out]), np.concatenate([o[3] for o in out]))
