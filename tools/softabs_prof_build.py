#!/usr/bin/env python3
"""Recompile ONLY k_softabs.hip - with the in-kernel phase clocks (-DMM_SOFTABS_PROF[=2]) or, with `off`, without them -
and relink both libraries (a full `python -m mici_amd.build --force` takes minutes on the build host).

    python tools/softabs_prof_build.py 1|2|off"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mici_amd import build as mb  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "1"
flags = [] if mode == "off" else ["-DMM_SOFTABS_PROF" + ("=2" if mode == "2" else "")]
src = os.path.join(mb.CSRC, "k_softabs.hip")
for objdir, extra in ((mb.OBJ, []), (mb.OBJ_DEV, ["-D" + mb.DEV_MACRO])):
    obj = os.path.join(objdir, "k_softabs.o")
    cmd = [mb.hipcc(), *mb.FLAGS, *extra, *flags, "-c", src, "-o", obj]
    subprocess.run(cmd, check=True)
mb.build(verbose=True)
