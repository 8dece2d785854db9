"""CPU suite (world_size 2, gloo): the N>1 host logic - contiguous chain shards, padded equal-length
batches, rank-major gather back to global chain order, per-rank independence of the integration -
checked against a single-process run.  The per-shard "integration" is done by the oracle here (no
GPU in this container); on the GPU box the same sharding feeds DeviceBatch + RcclTraceGather."""

import os
import socket

import numpy as np
import pytest

from mici_amd import distributed as mdist


def test_shard_bounds_and_padding():
    for n in (0, 1, 5, 16, 17, 4096):
        for w in (1, 2, 3, 8):
            b = mdist.shard_bounds(n, w)
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [s1 - s0 for s0, s1 in b]
            assert max(sizes) - min(sizes) <= 1
            x = np.arange(n * 3, dtype=np.float64).reshape(n, 3)
            shards = [mdist.take_shard(x, r, w) for r in range(w)]
            want = mdist.padded_shard_len(n, w)
            assert all(s.shape == (want, 3) for s in shards)
            if n:
                back = mdist.unpad_gathered(np.concatenate(shards), n, w)
                assert np.array_equal(back, x)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n, dim, steps, h, out_dir):
    import torch.distributed as dist

    from oracle import integrators as orc
    from oracle import models as omdl

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(42)  # same global inputs on every rank
    P = omdl.make_spd(dim, rng)
    q0, p0 = rng.standard_normal((2, n, dim))
    system = orc.EuclidSystem(omdl.GaussDense(P))
    ql = mdist.take_shard(q0, rank, world)
    pl = mdist.take_shard(p0, rank, world)
    qs, ps = orc.leapfrog_steps_batch(system, ql, pl, h, steps)  # this rank's shard only
    q_all = mdist.gather_host(qs, n)
    p_all = mdist.gather_host(ps, n)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), q=q_all, p=p_all)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [10, 7])
def test_two_rank_sharded_run_matches_single_process(tmp_path, n):
    import torch.multiprocessing as mp

    from oracle import integrators as orc
    from oracle import models as omdl

    dim, steps, h, world = 6, 5, 0.1, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, dim, steps, h, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(42)
    P = omdl.make_spd(dim, rng)
    q0, p0 = rng.standard_normal((2, n, dim))
    q_ref, p_ref = orc.leapfrog_steps_batch(orc.EuclidSystem(omdl.GaussDense(P)), q0, p0, h, steps)
    for r in range(world):
        got = np.load(tmp_path / f"rank{r}.npz")
        assert got["q"].shape == (n, dim)
        assert np.array_equal(got["q"], q_ref), f"rank {r}: gathered positions differ"
        assert np.array_equal(got["p"], p_ref)


# ---- the product's own rendezvous (standard library, no torch): mici_amd/rendezvous.py ------------------------
def _rdzv_worker(rank, world, path, n, dim, steps, h, out_dir):
    from mici_amd.rendezvous import Rendezvous
    from oracle import integrators as orc
    from oracle import models as omdl

    with Rendezvous(rank, world, path, timeout=60.0) as rdzv:
        rng = np.random.default_rng(42)
        P = omdl.make_spd(dim, rng)
        q0, p0 = rng.standard_normal((2, n, dim))
        system = orc.EuclidSystem(omdl.GaussDense(P))
        qs, ps = orc.leapfrog_steps_batch(system, mdist.take_shard(q0, rank, world), mdist.take_shard(p0, rank, world),
                                          h, steps)
        q_all = mdist.gather_host(qs, n, rdzv)
        p_all = mdist.gather_host(ps, n, rdzv)
        token = rdzv.broadcast(b"unique-id-from-rank-0" if rank == 0 else None)
        assert token == b"unique-id-from-rank-0"
        assert rdzv.reduce_max(float(rank)) == world - 1
        assert rdzv.reduce_sum(1.0) == world
        rdzv.barrier()
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), q=q_all, p=p_all)


@pytest.mark.parametrize("world,n", [(2, 10), (3, 7)])
def test_socket_rendezvous_sharded_run_matches_single_process(tmp_path, world, n):
    import multiprocessing as mp

    from oracle import integrators as orc
    from oracle import models as omdl

    dim, steps, h = 6, 5, 0.1
    path = str(tmp_path / "rdzv.sock")
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_rdzv_worker, args=(r, world, path, n, dim, steps, h, str(tmp_path)))
             for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    rng = np.random.default_rng(42)
    P = omdl.make_spd(dim, rng)
    q0, p0 = rng.standard_normal((2, n, dim))
    q_ref, p_ref = orc.leapfrog_steps_batch(orc.EuclidSystem(omdl.GaussDense(P)), q0, p0, h, steps)
    for r in range(world):
        got = np.load(tmp_path / f"rank{r}.npz")
        assert np.array_equal(got["q"], q_ref) and np.array_equal(got["p"], p_ref)


def test_bench_self_launch_reaches_rendezvous_and_fails_loudly_without_devices():
    """`python bench.py --gpus 2` with no WORLD_SIZE must start the ranks itself - and say so when the box has
    fewer than 2 devices (here: none) instead of silently measuring one GPU."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    # both ranks met at the rendezvous (each reports the count agreed there) and refused
    assert r.stderr.count("only 0 HIP device(s) are visible") == 2, r.stderr
    assert "rank exit codes [1, 1]" in r.stderr
    assert '"n_gpus"' not in r.stdout  # no result line was fabricated


def test_spawn_ranks_sets_rank_environment(tmp_path):
    import sys

    from mici_amd.rendezvous import spawn_ranks

    script = tmp_path / "w.py"
    script.write_text(
        "import os, sys\n"
        f"sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})\n"
        "from mici_amd.rendezvous import Rendezvous\n"
        "r = Rendezvous.from_env(timeout=60)\n"
        "got = r.allgather(os.environ['LOCAL_RANK'].encode())\n"
        "assert got == [b'0', b'1', b'2'], got\n"
        "r.barrier(); r.close()\n")
    assert spawn_ranks([sys.executable, str(script)], 3, timeout=120) == [0, 0, 0]
