T=$(mktemp -d); trap "rm -rf $T" EXIT  # scratch of THIS invocation (no fixed /tmp names)
# repeat the first tests of tests/test_gpu_user_target.py (an abort was seen twice in the sixth, never explained)
for i in $(seq 1 ${N:-10}); do
  python -X faulthandler -m pytest tests/test_gpu_user_target.py -q -x -k "builtin_target or new_target or compile_errors or torus_constraint" > $T/fl_$i.log 2>&1
  echo "run $i rc=$? $(tail -1 $T/fl_$i.log | cut -c1-100)"
  grep -q "Fatal" $T/fl_$i.log && { head -30 $T/fl_$i.log; dmesg 2>/dev/null | tail -5; }
done
