#!/bin/bash
# round 5: lock-step position solves on the matrix-core wave kernel (c3): variants built with tools/ab_build.py
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { # name env...
  local name=$1; shift
  for rep in 1 2; do
    env "$@" python bench.py --config c3 --no-extra-configs --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', '%.4e steps/s  %.3f ms/launch' % (d['value'], d['roofline']['kernel_ms_per_launch']))"
  done
}
{
run product_dual A=1
run product_dual_off MICI_AMD_DUAL=0
run nodual_build MICI_AMD_LIB=mici_amd/lib/ab_nodual.so
run dual_rsreg MICI_AMD_LIB=mici_amd/lib/ab_dual_rsreg.so
run dual_g1 MICI_AMD_LIB=mici_amd/lib/ab_dual_g1.so
} 2>&1 | tee gpurun_out/r05_ab_c3.txt
