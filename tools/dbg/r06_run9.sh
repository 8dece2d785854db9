cd $GRAFT_REPO_ROOT
MICI_AMD_FORK=1 timeout 900 python -m pytest tests/test_gpu_implicit.py -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r06_t9_tests.txt
cat gpurun_out/r06_t9_tests.txt
rm -f gpurun_out/r06_t9_ab.txt
for v in "FORK=1" "FORK=0" "FORK=1 MICI_AMD_DUAL=0"; do
  for i in 1 2; do
    env MICI_AMD_$v timeout 300 python bench.py --config c3 --no-extra-configs --no-cpu-baseline --steps 10 --warmup 2 2>&1 | tail -1 | python -c "
import sys,json
l=sys.stdin.read().strip()
try:
  d=json.loads(l); print('$v', d['value'], d['ms_per_step'])
except Exception as e: print('$v ERR', l[-400:])
" >> gpurun_out/r06_t9_ab.txt
  done
done
cat gpurun_out/r06_t9_ab.txt
