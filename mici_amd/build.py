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
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-fvisibility=hidden", "-Wall",
         "-Wno-unused-function", "-ffp-contract=on"]


# per-file code-generation switches
EXTRA_FLAGS = {
    # keep the MFMA accumulators (= the metric tiles) in architected VGPRs: the sweep reads them back every
    # block, and from AGPRs that is one v_accvgpr_read per dword
    "k_implicit_mfma.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
    "k_implicit_blk16.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
    "k_implicit_pair.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
    "k_implicit_fork.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
}


# development builds (in-kernel phase timers etc.): MICI_AMD_HIPCC_FLAGS="-DMM_SOFTABS_PROF" python -m mici_amd.build --force
DEV_FLAGS = os.environ.get("MICI_AMD_HIPCC_FLAGS", "").split()


HEADER = os.path.join(HERE, "..", "include", "mici_amd.h")


def exported_names(dev=False):
    """The functions include/mici_amd.h declares (the product's whole export list); dev: + the developer hooks."""
    import re
    with open(HEADER, encoding="utf-8") as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(mm_[a-z0-9_]+)\s*\(", text)))
    return names + (["mm_debug_*"] if dev else [])


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found (need ROCm to build libmici_amd.so)")
    return exe


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def _deps_time(path, seen=None):
    """Newest modification time of `path` and of everything it #includes with quotes, recursively (csrc/ and include/)."""
    import re
    seen = set() if seen is None else seen
    path = os.path.normpath(path)
    if path in seen or not os.path.exists(path):
        return 0.0
    seen.add(path)
    t = os.path.getmtime(path)
    with open(path, encoding="utf-8") as f:
        text = f.read()
    for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', text, flags=re.M):
        for base in (os.path.dirname(path), CSRC, os.path.join(HERE, "..", "include")):
            cand = os.path.join(base, inc)
            if os.path.exists(cand):
                t = max(t, _deps_time(cand, seen))
                break
    return t


def _compile_and_link(sources, objdir, lib, extra, headers_time, force, jobs, verbose, reuse=None):
    """Objects of `sources` into `objdir` (flags + extra), linked with the objects of `reuse` into `lib`.  An object is
    rebuilt when its source or a header it includes (recursively) is newer, or when this script is."""
    os.makedirs(objdir, exist_ok=True)
    cc = hipcc()
    todo, objs = [], []
    for src in sources:
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(_deps_time(src), headers_time):
            todo.append((src, obj))

    def compile_one(pair):
        src, obj = pair
        cmd = [cc, *FLAGS, *EXTRA_FLAGS.get(os.path.basename(src), []), *extra, *DEV_FLAGS, "-c", src, "-o", obj]
        import time
        t0 = time.time()
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode == 0:  # stamped with the time the compile STARTED: a header edited while it ran rebuilds it next time
            os.utime(obj, (t0, t0))
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
    if todo or force or not os.path.exists(lib) or os.path.getmtime(lib) < _newest(all_objs + [HEADER]):
        # the dynamic symbol table is exactly the export list: what include/mici_amd.h declares (+ mm_debug_* in the
        # developer library).  -fvisibility=hidden alone leaves the kernels' host-side handles, __hip_cuid_* and weak
        # std:: instantiations exported; a linker version script does not.
        vmap = os.path.join(objdir, "exports.map")
        with open(vmap, "w", encoding="utf-8") as f:
            f.write("{\n  global:\n" + "".join(f"    {n};\n" for n in exported_names(dev=bool(extra))) + "  local: *;\n};\n")
        cmd = [cc, f"--offload-arch={ARCH}", "-shared", "-fPIC", f"-Wl,--version-script={vmap}", *all_objs, "-o", lib, "-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stdout + r.stderr)
        if verbose:
            print(f"[mici_amd.build] linked {lib}")
    return objs


RTC_HEADERS = ["constrained_core.h", "constrained_wave.h", "implicit_wave.h", "implicit_mfma.h", "implicit_fork.h", "implicit_blk16.h", "implicit_team.h",
               "implicit_global.h",
               "softabs.h", "user_metric.h", "user_hessian.h", "implicit_core.h", "mm_device.h",
               os.path.join("..", "..", "include", "mici_amd.h")]
RTC_GEN = os.path.join(CSRC, "rtc_headers_gen.inc")


def gen_rtc_headers():
    """csrc/rtc_headers_gen.inc: the headers the run-time compiled translation units include (mm_rtc.hip hands them to
    hipRTC from memory - a deployed library does not need the source tree), as C++ raw string literals."""
    parts = ["// generated by mici_amd/build.py from " + ", ".join(os.path.basename(h) for h in RTC_HEADERS) + " - do not edit\n"]
    names, bodies = [], []
    for h in RTC_HEADERS:
        with open(os.path.join(CSRC, h), encoding="utf-8") as f:
            text = f.read()
        assert ')MMRTCH"' not in text
        names.append(os.path.basename(h))
        # a raw string literal may not exceed the compiler's limit comfortably: split into 8 KB chunks, concatenated
        chunks = [text[i:i + 8000] for i in range(0, len(text), 8000)] or [""]
        bodies.append("\n".join('R"MMRTCH(' + c + ')MMRTCH"' for c in chunks))
    parts.append("static const char* const kRtcHeaderNames[] = {" + ", ".join('"%s"' % n for n in names) + "};\n")
    parts.append("static const char* const kRtcHeaderSources[] = {\n" + ",\n".join(bodies) + "\n};\n")
    parts.append("static const int kRtcHeaderCount = %d;\n" % len(names))
    new = "".join(parts)
    old = None
    if os.path.exists(RTC_GEN):
        with open(RTC_GEN, encoding="utf-8") as f:
            old = f.read()
    if old != new:  # keep the time stamp when nothing changed: it is a dependency of every object
        with open(RTC_GEN, "w", encoding="utf-8") as f:
            f.write(new)


def build(force=False, jobs=None, verbose=True, dev=True):
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    gen_rtc_headers()
    sources = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdr_time = os.path.getmtime(os.path.abspath(__file__))  # the flags live here; headers are tracked per source
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
