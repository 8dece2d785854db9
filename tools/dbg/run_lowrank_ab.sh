# A/B of the Woodbury path: MICI_AMD_LOWRANK (solves) x MICI_AMD_LOWRANK_REFRESH (inverse updates in a row; 0 = sweep every step)
mkdir -p gpurun_out/lr
for cfg in c3 c4 c4_d512; do
  for v in "1 64" "1 0" "0 0"; do
    set -- $v
    MICI_AMD_LOWRANK=$1 MICI_AMD_LOWRANK_REFRESH=$2 timeout 600 python bench.py --config $cfg --no-cpu-baseline --no-extra-configs > gpurun_out/lr/bench_${cfg}_lr$1_rf$2.json 2> gpurun_out/lr/bench_${cfg}_lr$1_rf$2.err
    python - <<P
import json
try:
    d=json.loads(open("gpurun_out/lr/bench_${cfg}_lr$1_rf$2.json").read().strip().splitlines()[-1])
    print("${cfg} lowrank=$1 refresh=$2", d["value"], d["ms_per_step"], d["roofline"].get("frac"))
except Exception as e:
    print("${cfg} lr$1 rf$2 parse fail", e)
P
  done
done
cp bench_configs.json gpurun_out/lr/ 2>/dev/null
timeout 2400 python -m pytest tests/test_gpu_full_shards_all_chains.py tests/test_gpu_implicit.py tests/test_gpu_global_tier.py -q -m gpu -s -k "c3 or c4 or refined_solves or global" 2>&1 | grep -v Warning | tail -30 > gpurun_out/lr/tests_parity2.txt
cat gpurun_out/lr/tests_parity2.txt
