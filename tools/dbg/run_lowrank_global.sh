mkdir -p gpurun_out/lr
for v in "1 1" "1 0" "0 1"; do
  set -- $v
  MICI_AMD_LOWRANK=$1 MICI_AMD_DUAL=$2 timeout 600 python bench.py --config c4_d512 --no-cpu-baseline --no-extra-configs > gpurun_out/lr/bench_c4d512_lr$1_dual$2.json 2> gpurun_out/lr/bench_c4d512_lr$1_dual$2.err
  python - <<P
import json
d=json.loads(open("gpurun_out/lr/bench_c4d512_lr$1_dual$2.json").read().strip().splitlines()[-1])
print("c4_d512 lowrank=$1 dual=$2", d["value"], d["ms_per_step"])
P
done
timeout 1200 python -m pytest tests/test_gpu_global_tier.py -x -q -m gpu 2>&1 | tail -5
