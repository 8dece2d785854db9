mkdir -p gpurun_out/lr
timeout 2400 python -m pytest tests/test_gpu_full_shards_all_chains.py tests/test_gpu_implicit.py -q -m gpu -s -k "c3 or c4 or refined_solves" 2>&1 | tail -40 > gpurun_out/lr/tests_parity.txt
cat gpurun_out/lr/tests_parity.txt
