cd $GRAFT_REPO_ROOT
for n in ag12 default; do
  lib=mici_amd/lib/ab_fk_$n.so; [ $n = default ] && lib=mici_amd/lib/libmici_amd.so
  for i in 1 2; do
    MICI_AMD_LIB=$lib timeout 300 python bench.py --config c3 --no-extra-configs --no-cpu-baseline --steps 10 --warmup 2 2>&1 | tail -1 | python -c "
import sys,json
l=sys.stdin.read().strip()
try:
  d=json.loads(l); print('fk_$n', d['value'], d['ms_per_step'])
except Exception as e: print('fk_$n ERR', l[-300:])
"
  done
done
MICI_AMD_LIB=mici_amd/lib/ab_fk_ag12.so timeout 600 python -m pytest tests/test_gpu_fork.py -x -q -m gpu 2>&1 | tail -2
