#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02v; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "riemann or implicit or c3 or midpoint" > $O/pytest_c3.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_c3.log
timeout 300 python bench.py --config c3 --steps 10 --warmup 2 --no-cpu-baseline --no-extra-configs 2>/dev/null | grep -o '"value": [0-9.e+]*\|"work_counters": {[^}]*}' | head -2
