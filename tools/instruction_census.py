#!/usr/bin/env python3
"""Static instruction census of the kernels of one csrc/*.hip (device assembly from tools/kernel_asm.sh):
instructions by class, per kernel and per loop nest level.

    python tools/instruction_census.py k_constrained [--filter SpecTorus]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def census(lines):
    c = dict(total=0, valu=0, valu_f64=0, salu=0, branch=0, trans=0, div_expansion=0, vmem=0, smem=0, lds=0, mfma=0,
             exec_mask=0)
    for ln in lines:
        m = re.match(r"\s+([a-z_0-9]+)", ln)
        if not m:
            continue
        op = m.group(1)
        if not re.match(r"(v_|s_|ds_|global_|buffer_|scratch_|flat_)", op):
            continue
        c["total"] += 1
        if op.startswith("v_mfma"):
            c["mfma"] += 1
        elif op.startswith("v_"):
            c["valu"] += 1
            if "f64" in op:
                c["valu_f64"] += 1
            if re.match(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)", op):
                c["trans"] += 1
            if op.startswith("v_div_scale"):
                c["div_expansion"] += 1
        elif op.startswith("s_cbranch") or op == "s_branch":
            c["branch"] += 1
            c["salu"] += 1
        elif op.startswith("s_load") or op.startswith("s_buffer_load"):
            c["smem"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
            if "saveexec" in op or ("exec" in ln and op.startswith(("s_or_b64", "s_and_b64", "s_andn2", "s_xor", "s_mov_b64"))):
                c["exec_mask"] += 1
        elif op.startswith("ds_"):
            c["lds"] += 1
        else:
            c["vmem"] += 1
    return c


def main():
    args = sys.argv[1:]
    flt = None
    if "--filter" in args:
        i = args.index("--filter")
        flt = args[i + 1]
        del args[i:i + 2]
    name = args[0]
    extra = args[1:]
    out = subprocess.run([os.path.join(ROOT, "tools", "kernel_asm.sh"), name, *extra], capture_output=True, text=True)
    path = out.stdout.strip().splitlines()[-1]
    text = open(path).read().splitlines()
    starts = [(i, ln[:-1].split(":")[0]) for i, ln in enumerate(text) if re.match(r"^_Z\w+:", ln)]
    for k, (i, sym) in enumerate(starts):
        end = next((j for j in range(i, len(text)) if text[j].startswith("\t.end_amdhsa_kernel") or
                    text[j].startswith(".Lfunc_end")), len(text))
        dem = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip().split("(")[0]
        if flt and flt not in dem:
            continue
        body = text[i:end]
        c = census(body)
        print(f"{dem}")
        print("  whole kernel: " + "  ".join(f"{k}={v}" for k, v in c.items()))
        # loops: llvm annotates loop headers with '; ... Loop Header: Depth=N' / 'in Loop: ... Depth=N'
        depth_of_block, cur = {}, 0
        per_depth = {}
        for ln in body:
            m = re.search(r"Depth=(\d+)", ln)
            if re.match(r"^\.LBB", ln):
                cur = int(m.group(1)) if m else 0
            per_depth.setdefault(cur, []).append(ln)
        for d in sorted(per_depth):
            cd = census(per_depth[d])
            print(f"  blocks at loop depth {d}: total={cd['total']} valu={cd['valu']} (f64 {cd['valu_f64']}) salu={cd['salu']} "
                  f"branch={cd['branch']} trans={cd['trans']} div_expansions={cd['div_expansion']} exec_mask={cd['exec_mask']}")


if __name__ == "__main__":
    main()
