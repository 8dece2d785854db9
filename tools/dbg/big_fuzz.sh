T=$(mktemp -d); trap "rm -rf $T" EXIT  # scratch of THIS invocation (no fixed /tmp names)
# a longer randomised parity run than tools/refresh_profiles.sh affords (results: gpurun_out/big_fuzz.txt)
mkdir -p gpurun_out; O=gpurun_out/big_fuzz.txt; : > $O
run() { timeout 1500 python tools/fuzz_parity.py --seed $1 --cases $2 --kinds $3 $4 > $T/f_$1.log 2>&1; echo "seed $1 kinds $3 $4 rc=$? $(tail -1 $T/f_$1.log)" >> $O; grep "MISMATCH\|Traceback\|refused" -B3 $T/f_$1.log | head -20 >> $O; }
run 71 150 riemann_user &
run 72 200 softabs_user &
wait
run 73 300 riemann &
run 74 120 softabs --long &
wait
run 75 200 euclid,constrained
cat $O
