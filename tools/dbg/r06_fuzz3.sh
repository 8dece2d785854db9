cd $GRAFT_REPO_ROOT
O=gpurun_out/r06p; mkdir -p $O
for seed in 31 32 33; do
  timeout 1200 python tools/fuzz_parity.py --seed $seed --cases 80 > $O/fuzz_$seed.log 2>&1
  echo "seed $seed rc=$? $(tail -1 $O/fuzz_$seed.log)" >> $O/fuzz_parity_a.txt
  grep "MISMATCH\|Traceback" -B2 $O/fuzz_$seed.log | head -10 >> $O/fuzz_parity_a.txt
  rm -f $O/fuzz_$seed.log
done
cat $O/fuzz_parity_a.txt
