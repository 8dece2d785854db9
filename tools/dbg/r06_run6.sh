cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_global_tier.py tests/test_gpu_implicit.py -x -q -m gpu -k "global" 2>&1 | tail -3
MICI_AMD_DUAL=0 timeout 900 python -m pytest tests/test_gpu_global_tier.py -x -q -m gpu 2>&1 | tail -2
for d in 1 0; do
MICI_AMD_DUAL=$d timeout 500 python tools/time_global.py 2>&1 | tail -1
MICI_AMD_DUAL=$d timeout 600 python bench.py --config c4_d512 --no-extra-configs --no-cpu-baseline --steps 5 --warmup 1 2>&1 | tail -1 | cut -c1-120
done
