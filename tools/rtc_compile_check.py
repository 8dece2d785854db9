"""Compile the run-time translation units of the user-metric kernel families WITHOUT a device (libhiprtc works on the build
host; only loading the code object needs a GPU): catches compile errors in the headers' MM_RMETRIC_USER branches, times the
compiles and prints each kernel's register / scratch / LDS usage from the code object's metadata.

    python tools/rtc_compile_check.py [--dim 64] [--source SOFTPLUS_RANK1_FAST] [--families wave,mfma,team,blk16]"""
import argparse
import ctypes as C
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("MICI_AMD_RTC_CACHE", "off")
os.environ.setdefault("MICI_AMD_RTC_SEED", "off")

FAMS = {"wave": 0, "mfma": 1, "team": 2, "blk16": 3, "softabs": 4, "global": 5}


def resources(path):
    """kernel name -> (vgpr, agpr, sgpr, scratch bytes, lds bytes) from the code object's notes"""
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", path], capture_output=True, text=True).stdout
    res = {}
    for blk in out.split("- .agpr_count:")[1:]:
        def grab(key):
            m = re.search(r"\." + key + r":\s+(\S+)", blk)
            return m.group(1) if m else "?"
        agpr = re.match(r"\s*(\d+)", blk).group(1)
        res[grab("name")] = dict(vgpr=grab("vgpr_count"), agpr=agpr, sgpr=grab("sgpr_count"),
                                 scratch=grab("private_segment_fixed_size"), lds=grab("group_segment_fixed_size"),
                                 vgpr_spill=grab("vgpr_spill_count"), sgpr_spill=grab("sgpr_spill_count"))
    return res


def main():
    import user_sources
    from mici_amd import _ffi

    ap = argparse.ArgumentParser()
    ap.add_argument("--dim", type=int, default=64)
    ap.add_argument("--source", default="SOFTPLUS_RANK1_FAST")
    ap.add_argument("--families", default=None)
    a = ap.parse_args()
    lib = C.CDLL(_ffi.lib_path(dev=True))
    lib.mm_debug_rtc_compile.restype = C.c_long
    lib.mm_debug_rtc_compile.argtypes = [C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_char_p]
    lib.mm_last_error.restype = C.c_char_p
    lib.mm_last_error.argtypes = [C.c_void_p]
    src = getattr(user_sources, a.source).encode()
    fams = a.families.split(",") if a.families else (
        ["wave"] + (["mfma"] if a.dim > 32 else []) if a.dim <= 64 else ["team"] + (["blk16"] if 75 < a.dim <= 256 else []))
    rc_all = 0
    for fam in fams:
        with tempfile.NamedTemporaryFile(suffix=".hsaco") as f:
            t0 = time.time()
            n = lib.mm_debug_rtc_compile(a.dim, 4, FAMS[fam], src, f.name.encode())  # target 4: banana
            dt = time.time() - t0
            if n < 0:
                print(f"{fam}: rc={n}\n{lib.mm_last_error(None).decode()[:6000]}")
                rc_all = 1
                continue
            print(f"{fam}: dim {a.dim}, {a.source}: {n} bytes of code in {dt:.1f} s")
            for name, r in resources(f.name).items():
                print(f"    {name}: {r}")
    return rc_all


if __name__ == "__main__":
    sys.exit(main())
