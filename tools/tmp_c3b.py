import numpy as np, sys, os, json, subprocess
sys.path.insert(0, ".")
if len(sys.argv) > 1:
    import bench
    rng = np.random.default_rng(1234)
    w = bench.make_workload("c3b", 256, rng)
    integ = w["integ"]
    q, p, st, nd = integ.step_batch(w["q0"], w["p0"], np.ones(256, np.int8), n_steps=100)
    c = integ.last_counters
    np.savez(sys.argv[1], q=q, p=p, st=st, nd=nd)
    print({k: c[k] for k in ("n_eigh", "n_refine", "n_newton_iters", "n_fp_evals", "n_metric")}, "ok", int((st == 0).sum()), "chain0", st[0], nd[0])
else:
    for mode in ("1", "0"):
        env = dict(os.environ, MICI_AMD_REFINE=mode)
        subprocess.run([sys.executable, __file__, f"/tmp/r{mode}.npz"], env=env, check=True)
    a, b = np.load("/tmp/r1.npz"), np.load("/tmp/r0.npz")
    print("status equal", np.array_equal(a["st"], b["st"]), "n_done equal", np.array_equal(a["nd"], b["nd"]))
    ok = a["st"] == 0
    print("max |dq|", np.max(np.abs(a["q"][ok] - b["q"][ok])), "max |dp|", np.max(np.abs(a["p"][ok] - b["p"][ok])))
