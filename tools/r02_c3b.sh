#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02s; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -k "softabs or c3b or SoftAbs or riemannian" > $O/pytest_softabs.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_softabs.log
timeout 300 python bench.py --config c3b --steps 3 --warmup 1 --no-cpu-baseline --no-extra-configs > $O/bench_c3b.json 2> $O/bench_c3b.err; tail -2 $O/bench_c3b.err; cat $O/bench_c3b.json
