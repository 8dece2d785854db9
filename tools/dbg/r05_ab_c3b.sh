#!/bin/bash
# round 5, second pass: the lock step in its one-loop form with the paired M(x) v as a real loop of 4 / 8 columns-of-four a trip
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { local name=$1; shift
  for rep in 1 2; do
    env "$@" python bench.py --config c3 --no-extra-configs --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', '%.4e steps/s  %.3f ms/launch' % (d['value'], d['roofline']['kernel_ms_per_launch']))"
  done; }
{
run product A=1
run dual_unroll4 MICI_AMD_LIB=mici_amd/lib/ab_dual4.so
run dual_unroll4_lockstep_off MICI_AMD_LIB=mici_amd/lib/ab_dual4.so MICI_AMD_DUAL=0
run dual_unroll8 MICI_AMD_LIB=mici_amd/lib/ab_dual8.so
} 2>&1 | tee gpurun_out/r05_ab_c3b.txt
MICI_AMD_LIB=mici_amd/lib/ab_dual4.so timeout 600 python -m pytest tests/test_gpu_implicit.py -q -m gpu -x 2>&1 | tail -3 | tee -a gpurun_out/r05_ab_c3b.txt
