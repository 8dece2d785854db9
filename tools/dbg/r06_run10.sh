cd $GRAFT_REPO_ROOT
rm -f gpurun_out/r06_t10_ab.txt
for n in t0g0 t4g0 t0g1 t4g1; do
  for i in 1 2; do
    MICI_AMD_LIB=mici_amd/lib/ab_fork_$n.so MICI_AMD_FORK=1 timeout 300 python bench.py --config c3 --no-extra-configs --no-cpu-baseline --steps 10 --warmup 2 2>&1 | tail -1 | python -c "
import sys,json
l=sys.stdin.read().strip()
try:
  d=json.loads(l); print('fork_$n', d['value'], d['ms_per_step'])
except Exception as e: print('fork_$n ERR', l[-400:])
" >> gpurun_out/r06_t10_ab.txt
  done
done
MICI_AMD_FORK=0 timeout 300 python bench.py --config c3 --no-extra-configs --no-cpu-baseline --steps 10 --warmup 2 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip()); print('one-wave', d['value'], d['ms_per_step'])" >> gpurun_out/r06_t10_ab.txt
cat gpurun_out/r06_t10_ab.txt
MICI_AMD_LIB=mici_amd/lib/ab_fork_t4g1.so MICI_AMD_FORK=1 timeout 600 python -m pytest tests/test_gpu_implicit.py -x -q -m gpu -k "riemann" 2>&1 | tail -2
