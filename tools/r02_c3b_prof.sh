#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02s; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -k "softabs or c3b or SoftAbs or riemannian" > $O/pytest_softabs.log 2>&1; echo "pytest rc=$?"; grep -v "softabs prof" $O/pytest_softabs.log | tail -5
timeout 300 python bench.py --config c3b --steps 1 --warmup 0 --no-cpu-baseline --no-extra-configs > $O/prof_c3b.json 2> $O/prof_c3b.err; grep -h "softabs prof" $O/prof_c3b.json $O/prof_c3b.err | head -5; grep -o '"value": [0-9.e+]*' $O/prof_c3b.json | head -1
