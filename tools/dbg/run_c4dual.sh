mkdir -p gpurun_out/lr
for v in 1 0; do
  MICI_AMD_DUAL=$v timeout 600 python bench.py --config c4 --no-cpu-baseline --no-extra-configs > gpurun_out/lr/b7_c4_dual$v.json 2> gpurun_out/lr/b7_c4_dual$v.err
  python - <<P
import json
d=json.loads(open("gpurun_out/lr/b7_c4_dual$v.json").read().strip().splitlines()[-1])
print("c4 dual=$v", d["value"], d["ms_per_step"])
P
done
timeout 2400 python -m pytest tests/test_gpu_implicit.py tests/test_gpu_full_shards_all_chains.py tests/test_gpu_blk16.py -q -m gpu -x -k "not softabs and not c3b and not c5 and not c2 and not c3" 2>&1 | tail -4
