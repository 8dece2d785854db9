mkdir -p gpurun_out/lr
for cfg in c4 c4_user_lowrank; do
  timeout 600 python bench.py --config $cfg --no-cpu-baseline --no-extra-configs > gpurun_out/lr/b5_$cfg.json 2> gpurun_out/lr/b5_$cfg.err
  python - <<P
import json
d=json.loads(open("gpurun_out/lr/b5_$cfg.json").read().strip().splitlines()[-1])
print("$cfg", d["value"], d["ms_per_step"])
P
done
timeout 2400 python -m pytest tests/test_gpu_implicit.py tests/test_gpu_full_shards_all_chains.py tests/test_gpu_user_target.py tests/test_gpu_blk16.py -q -m gpu -x -k "not softabs and not c3b and not c5 and not c2" 2>&1 | tail -5
