#!/bin/bash
# Randomised HIP-vs-oracle parity sweep on the GPU box (tools/fuzz_parity.py), all case families, three seeds; the
# look-ahead c4 kernel on the Riemannian family.  Logs in gpurun_out/r02f/.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02f; mkdir -p $O
for seed in 21 22 23; do
  timeout 1200 python tools/fuzz_parity.py --seed $seed --cases 80 > $O/fuzz_$seed.log 2>&1; echo "seed $seed rc=$? $(tail -1 $O/fuzz_$seed.log)"
  grep "MISMATCH\|Traceback" -B2 $O/fuzz_$seed.log | head -10
done
MICI_AMD_IMPLICIT_KERNEL=blk16la timeout 600 python tools/fuzz_parity.py --seed 24 --cases 40 --kinds riemann > $O/fuzz_la.log 2>&1; echo "blk16la rc=$? $(tail -1 $O/fuzz_la.log)"
