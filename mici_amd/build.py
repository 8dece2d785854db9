"""Build libmici_amd.so for gfx950 with hipcc (in-tree; cross-compiles without a GPU).

    python -m mici_amd.build [--force] [--jobs N] [--no-dev]

Each csrc/*.hip is compiled to an object (in parallel), then linked into mici_amd/lib/libmici_amd.so - the product:
it exports exactly what include/mici_amd.h declares.  The developer kernels (phase timers, the linear algebra of a
backend on its own, micro-benchmarks: everything between `#ifdef MM_DEV_KERNELS` in csrc/) go into a second library,
mici_amd/lib/libmici_amd_dev.so = the product objects with the few sources that have developer code recompiled with
-DMM_DEV_KERNELS; tests/test_gpu_blk16.py and tools/ubench_*.py open it with `Context(dev=True)`.
Objects are rebuilt only when a source or header is newer."""

from __future__ import annotations

import argparse
import concurrent.futures as cf
import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_build")
OBJ_DEV = os.path.join(CSRC, "_build_dev")
LIB = os.path.join(HERE, "lib", "libmici_amd.so")
LIB_DEV = os.path.join(HERE, "lib", "libmici_amd_dev.so")
DEV_MACRO = "MM_DEV_KERNELS"
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-Wall",
         "-Wno-unused-function", "-ffp-contract=on"]


# per-file code-generation switches
EXTRA_FLAGS = {
    # keep the MFMA accumulators (= the metric tiles) in architected VGPRs: the sweep reads them back every
    # block, and from AGPRs that is one v_accvgpr_read per dword
    "k_implicit_mfma.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
    "k_implicit_blk16.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
}


# development builds (in-kernel phase timers etc.): MICI_AMD_HIPCC_FLAGS="-DMM_SOFTABS_PROF" python -m mici_amd.build --force
DEV_FLAGS = os.environ.get("MICI_AMD_HIPCC_FLAGS", "").split()


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found (need ROCm to build libmici_amd.so)")
    return exe


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def _compile_and_link(sources, objdir, lib, extra, headers_time, force, jobs, verbose, reuse=None):
    """Objects of `sources` into `objdir` (flags + extra), linked with the objects of `reuse` into `lib`."""
    os.makedirs(objdir, exist_ok=True)
    cc = hipcc()
    todo, objs = [], []
    for src in sources:
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), headers_time):
            todo.append((src, obj))

    def compile_one(pair):
        src, obj = pair
        cmd = [cc, *FLAGS, *EXTRA_FLAGS.get(os.path.basename(src), []), *extra, *DEV_FLAGS, "-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        return src, r.returncode, r.stdout + r.stderr

    if todo:
        with cf.ThreadPoolExecutor(max_workers=jobs or min(8, len(todo))) as ex:
            for src, rc, log in ex.map(compile_one, todo):
                if verbose:
                    print(f"[mici_amd.build] hipcc {' '.join(extra)} {os.path.basename(src)} -> rc={rc}")
                if rc != 0:
                    raise RuntimeError(f"hipcc failed on {src}:\n{log}")
                if verbose and log.strip():
                    print(log)
    all_objs = objs + list(reuse or [])
    if todo or force or not os.path.exists(lib) or os.path.getmtime(lib) < _newest(all_objs):
        cmd = [cc, f"--offload-arch={ARCH}", "-shared", "-fPIC", *all_objs, "-o", lib, "-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stdout + r.stderr)
        if verbose:
            print(f"[mici_amd.build] linked {lib}")
    return objs


def build(force=False, jobs=None, verbose=True, dev=True):
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    sources = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    headers = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.inc")) + glob.glob(
        os.path.join(HERE, "..", "include", "*.h"))
    hdr_time = _newest(headers)
    objs = _compile_and_link(sources, OBJ, LIB, [], hdr_time, force, jobs, verbose)
    if dev:
        def has_dev(path):
            with open(path, encoding="utf-8") as f:
                return DEV_MACRO in f.read()
        dev_sources = [s for s in sources if has_dev(s)]
        dev_names = {os.path.basename(s)[:-4] + ".o" for s in dev_sources}
        shared = [o for o in objs if os.path.basename(o) not in dev_names]
        _compile_and_link(dev_sources, OBJ_DEV, LIB_DEV, ["-D" + DEV_MACRO], hdr_time, force, jobs, verbose,
                          reuse=shared)
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=None)
    ap.add_argument("--no-dev", action="store_true", help="skip libmici_amd_dev.so (developer / test kernels)")
    a = ap.parse_args()
    build(force=a.force, jobs=a.jobs, dev=not a.no_dev)
    sys.exit(0)
