mkdir -p gpurun_out
timeout 2700 python -m pytest tests -q -m gpu -x 2>&1 | grep -v "Warning\|warnings.warn\|np.tanh\|^$" | tail -25 > gpurun_out/r06_gpu_tests_tail.txt
cat gpurun_out/r06_gpu_tests_tail.txt
