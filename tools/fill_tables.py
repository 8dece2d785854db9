#!/usr/bin/env python3
"""Fill the measured tables of DESIGN.md / README.md from profiles/<round>_bench_configs.json (the full record bench.py
writes beside its compact stdout line; rounds 1-4: <round>_bench_default.json) + the PMC files, between the
<!-- R05_TABLE --> / <!-- R05_README_TABLE --> markers (ROUND=r05), so the documents quote what the committed record holds."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = os.environ.get("ROUND", "r06")


def sci(x):
    e = 0
    while x >= 10:
        x /= 10
        e += 1
    sup = str(e).translate(str.maketrans("0123456789", "⁰¹²³⁴⁵⁶⁷⁸⁹"))
    return f"{x:.2f}·10{sup}"


def main():
    # round 5: the stdout line is the compact headline; the full per-config record is the sidecar, committed beside it
    side = os.path.join(ROOT, "profiles", f"{ROUND}_bench_configs.json")
    rec = json.load(open(side if os.path.exists(side) else os.path.join(ROOT, "profiles", f"{ROUND}_bench_default.json")))
    rows = [("c2", "c2(iii) D=128, 4096 chains, leapfrog", rec)]
    names = {"c2i": "c2(i) iso-Gaussian", "c2iv": "c2(iv) + dense metric", "c3": "c3(a) D=64, 1024 chains",
             "c3b": "c3(b) SoftAbs D=64", "c4": "c4 shard D=256, 1024 chains", "c5": "c5 shard, 2048 chains",
             "c3_user": "c3_user D=64: softplus + rank-one metric as user source", "c4_general": "c4_general D=256: the c4 metric as user source",
             "c4_user_lowrank": "c4_user_lowrank D=256: the c4 metric as user source that declares its structure (MM_USER_LOWRANK)",
             "c3b_dense": "c3b_dense D=64: SoftAbs on the banana, Hessian as user source (h = 0.01)",
             "c4_d512": "c4_d512 D=512, 256 chains: the c4 workload on the global-memory tier",
             "c2i_stream": "c2i_stream: c2(i) with n_steps = 1, 2²⁰ chains (the HBM-bound regime)",
             "c3b_d128": "c3b_d128: SoftAbs funnel D=128, 256 chains (workspace tier)",
             "c3b_d256": "c3b_d256: SoftAbs funnel D=256, 256 chains (workspace tier)"}
    for k, v in rec.get("configs", {}).items():
        if "error" not in v:
            rows.append((k, names.get(k, k), v))
    out = ["| config | steps/s (1 GPU) | kernel ms per launch | roofline (algorithmic; executed flops where the Woodbury path runs) | executed | HBM traffic per launch (PMC) |",
           "|---|---|---|---|---|---|"]
    for key, name, r in rows:
        roof = r["roofline"]
        frac = f"{roof['frac']:.3f} of {'FP64 MFMA' if roof['bound'] == 'mfma' else 'HBM'} peak ({roof['achieved']:.1f} {roof['unit']})"
        if "fp64_valu" in roof:
            frac += f"; FP64 VALU {roof['fp64_valu']['frac']:.2f}"
        if roof.get("reference_algorithm"):  # round 6: the Woodbury path - `frac` prices executed flops
            ra = roof["reference_algorithm"]
            frac = (f"executed: {roof['frac']:.3f} of FP64 peak ({roof['achieved']:.1f} TFLOP/s); the reference's algorithm at this "
                    f"rate would need {ra['ratio_to_fp64_peak']:.2f} × peak")
        ex = "—"
        if roof.get("hbm_model"):
            hm = roof["hbm_model"]
            ex = f"HBM-bound: modelled {hm['bytes_per_launch'] / 1e9:.0f} GB per launch = {hm['achieved_GBs'] / 1e3:.2f} TB/s ({hm['frac_of_hbm_peak']:.2f} of peak)"
            if roof.get("executed"):
                ex += f"; MFMA busy {roof['mfma_busy']:.3f}"
                e = roof["executed"]
                if e.get("lowrank_solves_per_chain_step"):
                    ex += (f"; {e['lowrank_solves_per_chain_step']:.1f} Woodbury solves + {e['inverse_updates_per_chain_step']:.2f} "
                           f"inverse updates + {e['sweeps_per_chain_step']:.2f} sweeps per step")
        elif roof.get("executed"):
            e = roof["executed"]
            if e.get("lowrank_solves_per_chain_step"):
                ex = (f"{e['lowrank_solves_per_chain_step']:.1f} Woodbury solves + {e['inverse_updates_per_chain_step']:.2f} inverse "
                      f"updates + {e['sweeps_per_chain_step']:.2f} sweeps per step; MFMA busy {roof['mfma_busy']:.3f}")
            elif "refine_pairs_per_chain_step" in e:
                ex = (f"MFMA busy {roof['mfma_busy']:.3f}; {e['refine_pairs_per_chain_step']:.1f} CG pairs + "
                      f"{e['sweeps_per_chain_step']:.2f} sweeps per step")
            else:  # SoftAbs: eigenvector refinement
                ex = (f"MFMA busy {roof['mfma_busy']:.3f}; {e['mfma_products_per_chain_step']:.0f} NP³ products, "
                      f"{e['refined_decompositions_per_chain_step']:.1f} refined decompositions + "
                      f"{e['jacobi_sweeps_per_chain_step']:.2f} Jacobi sweeps per step")
        tr = roof.get("traffic")
        out.append(f"| {name} | {sci(r['value'])} | {roof['kernel_ms_per_launch']:.3g} | {frac} | {ex} | "
                   f"{'—' if tr is None else '%.1f MB' % (tr / 1e6)} |")
    table = "\n".join(out)
    for fn, marker in (("DESIGN.md", f"{ROUND.upper()}_TABLE"), ("README.md", f"{ROUND.upper()}_README_TABLE")):
        p = os.path.join(ROOT, fn)
        s = open(p).read()
        block = f"<!-- {marker} -->\n{table}\n<!-- /{marker} -->"
        if f"<!-- /{marker} -->" in s:
            s = re.sub(rf"<!-- {marker} -->.*?<!-- /{marker} -->", lambda m: block, s, flags=re.S)
        else:
            s = s.replace(f"<!-- {marker} -->", block)
        open(p, "w").write(s)
    print(table)


if __name__ == "__main__":
    main()
