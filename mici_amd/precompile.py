"""Ahead-of-time compilation of user-metric kernels - no GPU needed.

    python -m mici_amd.precompile [--jobs N] [--source file.hip --dims 64,256]

The dense-Riemannian backends are compiled around a user's metric source by hipRTC (csrc/mm_rtc.hip); the matrix-core
kernels take 15-50 s each.  libhiprtc compiles without a device, so the code objects can be produced where the library is
built and shipped with it: they land in ``<directory of libmici_amd.so>/rtc_cache`` (named by the hash of everything that
went into them), which the library consults before its per-user cache (``MICI_AMD_RTC_CACHE``, default
``~/.cache/mici_amd/rtc``) and before compiling.  Without arguments the package's example sources
(``mici_amd/user_examples.py``) and the sources of the GPU tests are compiled for the dimensions the tests and ``bench.py``
use; ``--source`` compiles a text of your own.  Uses the developer library (``libmici_amd_dev.so``)."""

from __future__ import annotations

import argparse
import ctypes as C
import multiprocessing as mp
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
SEED_DIR = os.path.join(HERE, "lib", "rtc_cache")
FAMILIES = {"wave": 0, "mfma": 1, "team": 2, "blk16": 3, "softabs": 4, "global": 5}
TEST_DIMS = (4, 5, 6, 8, 16, 20, 27, 32, 40, 48, 64, 70, 100, 128, 130, 200, 256, 270, 300, 600)


def families_of(dim):
    """The kernel families mm_rtc_launch_riemann uses for a user metric of this size."""
    if dim <= 64:
        return ["wave"] + (["mfma"] if dim > 32 else [])
    if dim > 279:
        return ["global"]  # the global-memory tier: one family, every kernel of the model
    return ["team"] + (["blk16"] if 75 < dim <= 256 else [])


def _compile(job):
    name, src, dim, fam, target = job
    os.environ["MICI_AMD_RTC_CACHE"] = SEED_DIR  # rtc_code() stores what it compiles there
    from . import _ffi
    lib = C.CDLL(_ffi.lib_path(dev=True))
    lib.mm_debug_rtc_compile.restype = C.c_long
    lib.mm_debug_rtc_compile.argtypes = [C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_char_p]
    lib.mm_last_error.restype = C.c_char_p
    lib.mm_last_error.argtypes = [C.c_void_p]
    t0 = time.time()
    n = lib.mm_debug_rtc_compile(dim, target, FAMILIES[fam], src.encode(), None)
    msg = "" if n >= 0 else lib.mm_last_error(None).decode("utf-8", "replace")[:2000]
    return name, dim, fam, n, time.time() - t0, msg


def default_jobs():
    from . import user_examples as ue
    sources = {"RANK1_AS_USER_FLAT": (ue.RANK1_AS_USER_FLAT, TEST_DIMS),
               "SOFTPLUS_RANK1_FAST": (ue.SOFTPLUS_RANK1_FAST, tuple(d for d in TEST_DIMS if d <= 64)),
               # (its aux block is 2 D + 2 doubles of the 560 a source may ask for: D <= 279)
               "SOFTPLUS_RANK1_FAST_WIDE": (ue.SOFTPLUS_RANK1_FAST_WIDE, tuple(d for d in TEST_DIMS if 64 < d <= 279))}
    # round 6: metrics that declare their constant + rank-one structure (MM_USER_LOWRANK): bench.py c4_user_lowrank and
    # tests/test_gpu_user_target.py - the families with a Woodbury path, and the auxiliary family of each size
    lowrank_dims = (64, 130, 256, 320)
    sources["RANK1_AS_USER_LOWRANK"] = (ue.RANK1_AS_USER_LOWRANK, lowrank_dims)
    for d in (48,) + lowrank_dims:  # (48: the riemann_sinrank1_* fixtures of tests/golden)
        if d <= 256:
            sources[f"SIN_RANK1_LOWRANK_{d}"] = (ue.sin_rank1_lowrank(d), (d,))
    tests = os.path.join(HERE, "..", "tests")
    if os.path.exists(os.path.join(tests, "user_sources.py")):  # the plain-form sources of the GPU tests
        sys.path.insert(0, tests)
        import user_sources as us
        sources["RANK1_AS_USER"] = (us.RANK1_AS_USER, TEST_DIMS)
        sources["SOFTPLUS_RANK1"] = (us.SOFTPLUS_RANK1, TEST_DIMS)
    jobs = []
    for name, (src, dims) in sources.items():
        for dim in dims:
            for fam in families_of(dim):
                jobs.append((name, src, dim, fam, 0))
    # a user TARGET joined with a user metric / Hessian (tests/test_gpu_user_target.py: the text the package hands over is
    # target source + "\n" + metric source, mici_amd/runtime.py DeviceModel)
    if "user_sources" in sys.modules:
        us = sys.modules["user_sources"]
        for msrc, dims in ((us.RANK1_AS_USER, (20, 70, 200)), (ue.RANK1_AS_USER_FLAT, (48, 100))):
            for dim in dims:
                for fam in families_of(dim):
                    jobs.append(("BANANA_SRC+RANK1", us.BANANA_SRC + "\n" + msrc, dim, fam, 100))
        jobs.append(("BANANA_SRC+BANANA_HESS", us.BANANA_SRC + "\n" + ue.BANANA_HESS, 64, "softabs", 100))
    # user Hessians of SoftAbs systems (one translation unit per text and padded size: dim <= 64, <= 128, <= 256)
    for dim in (64, 128, 256):
        jobs.append(("BANANA_HESS", ue.BANANA_HESS, dim, "softabs", 0))
        if "user_sources" in sys.modules:
            jobs.append(("FUNNEL_HESS", sys.modules["user_sources"].FUNNEL_HESS, dim, "softabs", 0))
    # slowest first: the matrix-core wave kernel, then the wave kernel at its largest tile size
    order = {"mfma": 0, "softabs": 0, "wave": 1, "blk16": 2, "team": 3, "global": 3}
    jobs.sort(key=lambda j: (order[j[3]], -j[2]))
    return jobs


def precompile(jobs, n_proc=None, verbose=True, clean=False):
    os.makedirs(SEED_DIR, exist_ok=True)
    if clean:  # code objects are named by the hash of the text AND the library's headers: an older build's are dead weight
        for f in os.listdir(SEED_DIR):
            if f.endswith(".hsaco"):
                os.remove(os.path.join(SEED_DIR, f))
    n_proc = n_proc or min(len(jobs), os.cpu_count() or 1)
    t0 = time.time()
    failed = 0
    with mp.get_context("spawn").Pool(n_proc) as pool:
        for name, dim, fam, n, dt, msg in pool.imap_unordered(_compile, jobs, chunksize=1):
            if n < 0:
                failed += 1
                print(f"[mici_amd.precompile] {name} dim {dim} {fam}: rc={n}\n{msg}", file=sys.stderr)
            elif verbose and dt > 1.0:
                print(f"[mici_amd.precompile] {name} dim {dim} {fam}: {n} bytes in {dt:.1f} s")
    if verbose:
        print(f"[mici_amd.precompile] {len(jobs)} translation units, {len(os.listdir(SEED_DIR))} code objects in {SEED_DIR} "
              f"({time.time() - t0:.0f} s)")
    return failed


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--jobs", type=int, default=None)
    ap.add_argument("--source", default=None, help="a HIP text defining mm_user_metric / mm_user_vjp[_flat]")
    ap.add_argument("--dims", default="64")
    ap.add_argument("--user-target", action="store_true", help="the text also defines the target (MM_TARGET_USER)")
    a = ap.parse_args()
    if a.source:
        with open(a.source, encoding="utf-8") as f:
            src = f.read()
        jobs = [(os.path.basename(a.source), src, int(d), fam, 100 if a.user_target else 0)
                for d in a.dims.split(",") for fam in families_of(int(d))]
    else:
        jobs = default_jobs()
    sys.exit(1 if precompile(jobs, a.jobs, clean=not a.source) else 0)


if __name__ == "__main__":
    main()
