#!/usr/bin/env python3
"""Register / scratch / LDS usage of every kernel of a csrc/*.hip file, as the compiler reports it
(-Rpass-analysis=kernel-resource-usage), compiled with the flags of mici_amd/build.py.

    python tools/kernel_resources.py k_implicit_blk16 [k_implicit_mfma ...] [--filter implicit_]"""
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mici_amd import build as mb  # noqa: E402


def report(name, flt):
    src = os.path.join(mb.CSRC, name if name.endswith(".hip") else name + ".hip")
    cmd = [mb.hipcc(), *mb.FLAGS, *mb.EXTRA_FLAGS.get(os.path.basename(src), []), *mb.DEV_FLAGS,
           "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur = None
    rows = {}
    for line in out.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            cur = re.sub(r"\(anonymous namespace\)::", "", cur).split("(")[0]
            rows[cur] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z][A-Za-z ]*?)\s*(?:\[[^\]]*\])?: (\S+) \[-Rpass", line)
        if m and cur:
            rows[cur][m.group(1).strip()] = m.group(2)
    for k, v in rows.items():
        if flt and flt not in k:
            continue
        print(f"{k}: VGPR {v.get('VGPRs')} AGPR {v.get('AGPRs')} SGPR {v.get('TotalSGPRs')} scratch "
              f"{v.get('ScratchSize')} B/lane occupancy {v.get('Occupancy')} sgpr-spill {v.get('SGPRs Spill')} "
              f"vgpr-spill {v.get('VGPRs Spill')}")


if __name__ == "__main__":
    args = sys.argv[1:]
    flt = None
    if "--filter" in args:
        i = args.index("--filter")
        flt = args[i + 1]
        del args[i:i + 2]
    for a in args:
        report(a, flt)
