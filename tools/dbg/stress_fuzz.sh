T=$(mktemp -d); trap "rm -rf $T" EXIT  # scratch of THIS invocation (no fixed /tmp names)
mkdir -p gpurun_out; O=gpurun_out/stress_fuzz.txt; : > $O
run() { timeout 1500 python tools/fuzz_parity.py --seed $1 --cases $2 --kinds $3 --stress $4 > $T/f_$1.log 2>&1; echo "seed $1 kinds $3 stress $4 rc=$? $(tail -1 $T/f_$1.log)" >> $O; grep "MISMATCH" -B3 $T/f_$1.log | head -40 >> $O;  }
run 81 100 riemann 10 &
run 82 100 softabs 8 &
run 83 60 riemann_user 10 &
run 84 80 softabs_user 6 &
wait
cat $O
