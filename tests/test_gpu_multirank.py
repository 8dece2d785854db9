"""N > 1 rank processes on the hardware there is: `bench.py --gpus 2` with both ranks on device 0.

RCCL refuses two ranks on one device, so the run uses the explicit test switch MICI_AMD_SHARE_DEVICE=1 (bench.py
docstring): one process per rank with its own context, the Unix-socket rendezvous, a rank-independent model and
rank-dependent chains, the trace gathered through the host rendezvous.  Role in the reference: the per-chain worker
processes and their trace collection (samplers.py:546-565, 668-772).  No scaling figure is read from this."""

import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run_bench(tmp_path, world, config, n_local, traj, steps, warmup):
    dump = os.path.join(str(tmp_path), f"trace_{config}_{world}.npy")
    env = dict(os.environ, MICI_AMD_SHARE_DEVICE="1", MICI_AMD_BENCH_DUMP_TRACE=dump, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--config", config, "--steps", str(steps),
           "--warmup", str(warmup), "--chains-per-gpu", str(n_local), "--traj-len", str(traj), "--no-cpu-baseline",
           "--no-extra-configs"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]  # rank 0 prints the one line
    return json.loads(lines[0]), np.load(dump)


@pytest.mark.parametrize("config,n_local,traj", [("c2", 96, 7), ("c3", 24, 2), ("c5", 200, 5)])
def test_two_ranks_on_one_device_reproduce_the_single_process_run(tmp_path, config, n_local, traj):
    import bench

    world, steps, warmup = 2, 2, 1
    line, gathered = _run_bench(tmp_path, world, config, n_local, traj, steps, warmup)
    assert line["n_gpus"] == world and line["steps"] == steps
    assert len(line["rank_elapsed_s"]["per_rank"]) == world
    assert line["rank_elapsed_s"]["max"] == max(line["rank_elapsed_s"]["per_rank"])
    assert "host gather" in line["config"]["trace_gather"] and line["config"]["trace_gather_ms"] > 0
    assert "MICI_AMD_SHARE_DEVICE" in line["config"]["parallelism"]
    assert gathered.shape[0] == world * n_local
    # the same chains in THIS process: rank r's inputs (the model from the rank-independent stream, the chains from
    # the rank's own), `steps` passes of `traj` steps each from the initial state - chain for chain, bit for bit
    total_done = 0
    for r in range(world):
        w = bench.make_workload(config, n_local, np.random.default_rng(1234),
                                chain_rng=None if r == 0 else np.random.default_rng([1234, r]))
        q, p = w["q0"], w["p0"]
        for _ in range(steps):
            q, p, st, nd = w["integ"].step_batch(q, p, 1, n_steps=traj)
            total_done += int(nd.sum())  # (what bench.py counts: completed steps of every pass)
        assert np.array_equal(gathered[r * n_local:(r + 1) * n_local], q), f"rank {r} shard differs"
    assert abs(line["value"] * line["rank_elapsed_s"]["max"] - total_done) <= 1e-6 * total_done


def test_share_switch_is_needed_on_a_one_gpu_box(tmp_path):
    """Without the switch a 2-rank launch on fewer than two devices still refuses (no silent oversubscription)."""
    import ctypes as C

    from mici_amd import _ffi
    count = C.c_int(0)
    assert _ffi.load().mm_device_count(C.byref(count)) == 0
    if count.value >= 2:
        pytest.skip("box has two devices")
    env = dict(os.environ)
    for k in ("MICI_AMD_SHARE_DEVICE", "WORLD_SIZE", "RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--no-cpu-baseline", "--no-extra-configs"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert "only 1 HIP device" in out.stderr
