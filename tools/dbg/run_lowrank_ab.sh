set -x
mkdir -p gpurun_out/lr
for cfg in c3 c4 c4_d512; do
  for lr in 1 0; do
    MICI_AMD_LOWRANK=$lr timeout 600 python bench.py --config $cfg --no-cpu-baseline --no-extra-configs > gpurun_out/lr/bench_${cfg}_lr${lr}.json 2> gpurun_out/lr/bench_${cfg}_lr${lr}.err
    python - <<P
import json
try:
    d=json.loads(open("gpurun_out/lr/bench_${cfg}_lr${lr}.json").read().strip().splitlines()[-1])
    print("${cfg} lowrank=${lr}", d["value"], d["ms_per_step"], d.get("roofline",{}).get("work_counters"))
except Exception as e:
    print("${cfg} lr${lr} parse fail", e)
P
  done
done
timeout 1500 python -m pytest tests/test_gpu_implicit.py tests/test_gpu_global_tier.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/lr/tests_implicit.txt
cat gpurun_out/lr/tests_implicit.txt
