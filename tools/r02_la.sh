#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02w; mkdir -p $O
export MICI_AMD_IMPLICIT_KERNEL=blk16la
timeout 900 python -m pytest tests -m gpu -x -q -k "riemann or c4 or blk16 or d100 or d70 or d256" > $O/pytest_la2.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_la2.log
timeout 300 python bench.py --config c4 --steps 4 --warmup 1 --no-cpu-baseline --no-extra-configs 2>/dev/null | grep -o '"value": [0-9.e+]*' | head -1
unset MICI_AMD_IMPLICIT_KERNEL
python tools/ubench_blk16la.py 2>&1 | tail -26
