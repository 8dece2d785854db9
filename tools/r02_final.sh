#!/bin/bash
# Round-2 closing run on the GPU box: the whole GPU test suite, the default bench line, rocprofv3 kernel stats of the
# same command, the construction timings of the two c4 kernels.  Results in gpurun_out/r02z/.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02z; mkdir -p $O
SECONDS=0
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$? (${SECONDS}s)"; tail -3 $O/pytest_gpu.log
SECONDS=0
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench wall ${SECONDS}s"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_default -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof_default.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/prof_default -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/rocprofv3_kernel_stats.csv && head -8 $O/rocprofv3_kernel_stats.csv | cut -c1-150
rm -rf $O/prof_default
python - <<PY
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print("c2 %.4g"%d["value"], "frac %.3f"%d["roofline"]["frac"], d["roofline"]["traffic"], "cpu %.4g"%d["cpu_baseline"]["value"])
for k,v in d["configs"].items():
    cb=v["cpu_baseline"]
    print(k, "%.4g"%v["value"], "frac %.3f"%v["roofline"]["frac"], "ms %.3f"%v["roofline"]["kernel_ms_per_launch"], "traffic", v["roofline"]["traffic"], "| cpu", "%.4g"%cb["value"])
PY
