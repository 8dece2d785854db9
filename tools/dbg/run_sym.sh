mkdir -p gpurun_out/lr
for v in 1 0; do
  MICI_AMD_GLOBAL_SYM=$v timeout 600 python bench.py --config c4_d512 --no-cpu-baseline --no-extra-configs > gpurun_out/lr/b4_sym$v.json 2> gpurun_out/lr/b4_sym$v.err
  python - <<P
import json
d=json.loads(open("gpurun_out/lr/b4_sym$v.json").read().strip().splitlines()[-1])
print("c4_d512 sym=$v", d["value"], d["ms_per_step"])
P
done
timeout 1500 python -m pytest tests/test_gpu_global_tier.py tests/test_gpu_implicit.py -q -m gpu -x -k "global or inverse_updates or riemann_global or fixture" 2>&1 | tail -5
