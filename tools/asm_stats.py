#!/usr/bin/env python3
"""Instruction census of the kernels in a device assembly file (tools/kernel_asm.sh output):
    python tools/asm_stats.py /tmp/k_implicit_mfma.s [name-filter]"""
import re
import sys

text = open(sys.argv[1]).read().split("\n")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cur, stats = None, {}
for ln in text:
    m = re.match(r"^(_Z\w+):\s", ln)
    if m:
        cur = m.group(1)
        stats[cur] = {}
        continue
    if cur is None:
        continue
    t = ln.strip().split(" ")[0]
    if not t or t.startswith((";", ".")) or t.endswith(":"):
        continue
    for key in ("v_accvgpr_read", "v_accvgpr_write", "v_mfma", "scratch_", "v_fma_f64", "ds_read", "ds_write", "ds_load", "ds_store",
                "s_waitcnt", "v_readlane", "v_mov_b32", "s_barrier"):
        if t.startswith(key):
            stats[cur][key] = stats[cur].get(key, 0) + 1
    stats[cur]["total"] = stats[cur].get("total", 0) + 1
    if t == "s_endpgm":
        cur = None
for k, v in stats.items():
    if flt in k and v.get("total", 0) > 50:
        print(k[:70], v)
