#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02f; mkdir -p $O
timeout 900 python tools/fuzz_parity.py --seed 11 --cases 60 --kinds softabs,constrained > $O/fuzz_sc.log 2>&1; echo "rc=$?"; tail -2 $O/fuzz_sc.log; grep -c "refused" $O/fuzz_sc.log; grep "MISMATCH\|Error\|Traceback" -B2 $O/fuzz_sc.log | head -20
timeout 900 python tools/fuzz_parity.py --seed 12 --cases 50 --kinds riemann,euclid > $O/fuzz_re.log 2>&1; echo "rc=$?"; tail -2 $O/fuzz_re.log; grep "MISMATCH\|Error\|Traceback" -B2 $O/fuzz_re.log | head -20
MICI_AMD_IMPLICIT_KERNEL=blk16la timeout 600 python tools/fuzz_parity.py --seed 13 --cases 30 --kinds riemann > $O/fuzz_la.log 2>&1; echo "rc=$?"; tail -1 $O/fuzz_la.log; grep "MISMATCH" -B2 $O/fuzz_la.log | head
