cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_euclid.py tests/test_gpu_chain_steps.py tests/test_gpu_gaussian.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2; do
timeout 600 python bench.py --config c2i_stream --no-extra-configs --no-cpu-baseline --steps 20 --warmup 3 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('c2i_stream', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'])"
done
