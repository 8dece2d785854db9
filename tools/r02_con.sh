#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02u; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "constrained or unsupported" > $O/pytest_con.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_con.log
