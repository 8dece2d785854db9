cd $GRAFT_REPO_ROOT
rm -f gpurun_out/all_chains.txt
timeout 900 python -m pytest tests/test_gpu_global_tier.py tests/test_gpu_implicit.py -x -q -m gpu -k "global" 2>&1 | tail -5 > gpurun_out/r06_t3_global_tests.txt
cat gpurun_out/r06_t3_global_tests.txt
for c in c4_d512 c2i_stream; do
  timeout 600 python bench.py --config $c --no-extra-configs --no-cpu-baseline --steps 5 --warmup 1 2>&1 | tail -1 | python -c "
import sys,json
l=sys.stdin.read().strip()
try:
  d=json.loads(l); print('$c', d['value'], d['ms_per_step'], json.dumps(d['roofline'])[:400])
except Exception as e: print('$c ERR', l[-600:])
" >> gpurun_out/r06_t3_bench.txt
done
cat gpurun_out/r06_t3_bench.txt
timeout 1500 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_full_shards_all_chains.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r06_t3_newtests.txt
cat gpurun_out/r06_t3_newtests.txt gpurun_out/all_chains.txt
