cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_implicit.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r06_t2_tests.txt
rm -f gpurun_out/r06_t2_ab.txt
for v in 1 0; do
  for i in 1 2; do
    MICI_AMD_PAIR=$v timeout 300 python bench.py --config c3 --no-extra-configs --no-cpu-baseline --steps 10 --warmup 2 2>&1 | tail -1 | python -c "
import sys,json
l=sys.stdin.read().strip()
try:
  d=json.loads(l); print('pair=$v', d['value'], d['ms_per_step'])
except Exception as e: print('pair=$v ERR', l[:300])
" >> gpurun_out/r06_t2_ab.txt
  done
done
timeout 300 python tools/ubench_pair.py > gpurun_out/r06_t2_prof.txt 2>&1
cat gpurun_out/r06_t2_tests.txt gpurun_out/r06_t2_ab.txt gpurun_out/r06_t2_prof.txt
