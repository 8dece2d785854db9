cd $GRAFT_REPO_ROOT
rm -f gpurun_out/all_chains.txt
timeout 1700 python -m pytest tests -q -m gpu -x 2>&1 | tail -6 > gpurun_out/r06_t11_tests.txt
cat gpurun_out/r06_t11_tests.txt
for i in 1 2; do timeout 300 python bench.py --config c3 --no-extra-configs --no-cpu-baseline --steps 20 --warmup 3 2>&1 | tail -1 | cut -c1-100; done
