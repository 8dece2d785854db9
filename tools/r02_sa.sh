#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02u; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "softabs or unsupported or riemannian or c3b" > $O/pytest_sa.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_sa.log
timeout 300 python bench.py --config c3b --steps 3 --warmup 1 --no-cpu-baseline --no-extra-configs 2>/dev/null | grep -o '"value": [0-9.e+]*' | head -1
