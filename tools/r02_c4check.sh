#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02y; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "riemann or c4 or blk16 or d100 or d70 or d256" > $O/pytest_c4.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_c4.log
timeout 300 python bench.py --config c4 --steps 6 --warmup 1 --no-cpu-baseline --no-extra-configs 2>/dev/null | grep -o '"value": [0-9.e+]*' | head -1
python tools/ubench_blk16.py 256 256 2>&1 | sed -n 1,5p; python tools/ubench_blk16.py 256 256 2>&1 | grep -A10 "^trailing sweep"
