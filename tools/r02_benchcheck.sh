#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02q; mkdir -p $O
SECONDS=0
python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench wall: ${SECONDS}s"; tail -3 $O/bench_default.err
python - <<PY
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print("c2 %.4g"%d["value"], "frac %.3f"%d["roofline"]["frac"], d["roofline"]["traffic"], "cpu %.4g"%d["cpu_baseline"]["value"])
for k,v in d["configs"].items():
    cb=v["cpu_baseline"]
    print(k, "%.4g"%v["value"], "frac %.3f"%v["roofline"]["frac"], "traffic", v["roofline"]["traffic"], "| cpu", cb["value"], cb.get("single_chain_1core"), cb["sample"][:90])
PY
