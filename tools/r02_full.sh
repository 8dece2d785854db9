#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02t; mkdir -p $O
SECONDS=0
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$? (${SECONDS}s)"; tail -3 $O/pytest_gpu.log
bash tools/r02_benchcheck.sh
cp gpurun_out/r02q/bench_default.json $O/ 2>/dev/null
