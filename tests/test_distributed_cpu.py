"""CPU suite (world_size 2, gloo): the N>1 host logic - contiguous chain shards, padded equal-length
batches, rank-major gather back to global chain order, per-rank independence of the integration -
checked against a single-process run.  The per-shard "integration" is done by the oracle here (no
GPU in this container); on the GPU box the same sharding feeds DeviceBatch + RcclTraceGather."""

import os
import socket

import numpy as np
import pytest

from mici_amd import distributed as mdist


def test_product_does_not_import_torch():
    """north_star: no PyTorch in the product.  The only torch in this repository is this file's gloo test."""
    import subprocess
    import sys
    code = ("import sys; import mici_amd; from mici_amd import distributed, rendezvous, transitions, adapters, "
            "integrators, systems, runtime, traces, interop; assert 'torch' not in sys.modules, 'torch imported'")
    subprocess.run([sys.executable, "-c", code], check=True,
                   cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    src = open(mdist.__file__).read()
    assert "import torch" not in src
    with pytest.raises(TypeError):
        mdist.gather_host(np.zeros((2, 2)), 2, None)


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


class TorchGroup:
    """The four members mici_amd.distributed asks of a group, on a torch.distributed (gloo) process group: the
    product imports no torch, a caller that lives inside a torch job brings this adapter."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self._dist, self._group = dist, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)

    def allgather_array(self, local):
        import torch
        t = torch.from_numpy(np.ascontiguousarray(local))
        out = [torch.empty_like(t) for _ in range(self.world)]
        self._dist.all_gather(out, t, group=self._group)
        return np.stack([o.numpy() for o in out])

    def broadcast(self, blob):
        import torch
        n = torch.tensor([len(blob) if self.rank == 0 else 0], dtype=torch.int64)
        self._dist.broadcast(n, src=0, group=self._group)
        buf = torch.tensor(list(blob), dtype=torch.uint8) if self.rank == 0 else torch.zeros(int(n), dtype=torch.uint8)
        self._dist.broadcast(buf, src=0, group=self._group)
        return bytes(buf.tolist())


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
    group = TorchGroup()
    q_all = mdist.gather_host(qs, n, group)
    p_all = mdist.gather_host(ps, n, group)
    blob = group.broadcast(b"id-from-rank-0" if rank == 0 else None)  # what exchange_unique_id sends
    assert blob == b"id-from-rank-0"
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


@pytest.mark.parametrize("world,n", [(2, 10), (3, 7), (8, 37)])  # (8: the node run of SURVEY 8e - VERDICT r05 #5 / #8)
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


def test_rendezvous_framing_and_private_directory():
    """no pickle on the wire (a frame is bytes + a dtype/shape header), 64-bit frame lengths, the default
    socket inside a directory only this user can enter."""
    import stat

    from mici_amd import rendezvous as rz

    src = open(rz.__file__).read()
    assert "import pickle" not in src and "pickle." not in src
    for a in (np.arange(12.0).reshape(3, 4), np.zeros((0, 5)), np.arange(6, dtype=np.int8).reshape(1, 2, 3),
              np.float64(3.5)):
        b = rz.unpack_array(rz.pack_array(a))
        assert b.dtype == np.asarray(a).dtype and b.shape == np.asarray(a).shape and np.array_equal(b, a)
    with pytest.raises(TypeError):
        rz.pack_array(np.array([object()]))
    parts = [b"", b"abc", bytes(range(256)) * 3]
    assert rz._unpack_parts(rz._pack_parts(parts)) == parts
    assert rz._HDR.size == 8
    d = rz.private_dir()
    st = os.lstat(d)
    assert st.st_uid == os.getuid() and stat.S_IMODE(st.st_mode) == 0o700
    old = os.environ.pop("MICI_AMD_RDZV", None)
    try:
        assert os.path.dirname(rz.default_path()) == d
    finally:
        if old is not None:
            os.environ["MICI_AMD_RDZV"] = old


def _tagged_worker(rank, world, path, tag, out_dir, expect_ok):
    from mici_amd.rendezvous import Rendezvous
    try:
        with Rendezvous(rank, world, path, timeout=6.0, tag=tag) as rdzv:
            got = rdzv.allgather(b"r%d" % rank)
        ok = got == [b"r%d" % r for r in range(world)]
    except (ConnectionError, TimeoutError):
        ok = False
    open(os.path.join(out_dir, f"res_{tag}_{rank}_{int(ok)}"), "w").close()
    raise SystemExit(0 if ok == expect_ok else 1)


def test_rendezvous_rejects_foreign_and_duplicate_ranks(tmp_path):
    """A rank of ANOTHER job (different tag) or a second claimant of a rank id that connects to this job's socket is
    turned away; the job's own ranks still complete their collective."""
    import multiprocessing as mp

    path = str(tmp_path / "rdzv.sock")
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_tagged_worker, args=(0, 2, path, "jobA", str(tmp_path), True)),
             ctx.Process(target=_tagged_worker, args=(1, 2, path, "jobB", str(tmp_path), False))]  # foreign job
    for p in procs:
        p.start()
    import time
    time.sleep(1.0)
    late = ctx.Process(target=_tagged_worker, args=(1, 2, path, "jobA", str(tmp_path), True))  # the real rank 1
    late.start()
    for p in procs + [late]:
        p.join(60)
        assert p.exitcode == 0
    assert (tmp_path / "res_jobA_0_1").exists() and (tmp_path / "res_jobA_1_1").exists()
    assert (tmp_path / "res_jobB_1_0").exists()


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
