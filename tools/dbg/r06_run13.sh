cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_implicit.py tests/test_gpu_user_target.py -x -q -m gpu -k "softabs or hessian" 2>&1 | tail -3
for c in c3b_d128 c3b_d256; do
timeout 600 python bench.py --config $c --no-extra-configs --no-cpu-baseline --steps 3 --warmup 1 2>&1 | tail -1 | python -c "
import sys,json
l=sys.stdin.read().strip()
try:
  d=json.loads(l); print('$c', d['value'], d['ms_per_step'], d['roofline']['frac'])
except Exception as e: print('$c ERR', l[-600:])
"
done
