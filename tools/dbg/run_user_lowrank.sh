mkdir -p gpurun_out/lr
timeout 2400 python -m pytest tests/test_gpu_user_target.py tests/test_gpu_fork.py -q -m gpu 2>&1 | grep -v "Warning\|np.tanh" | tail -40 > gpurun_out/lr/tests_user.txt
cat gpurun_out/lr/tests_user.txt
for cfg in c4_user_lowrank c4_general; do
timeout 900 python bench.py --config $cfg --no-cpu-baseline --no-extra-configs > gpurun_out/lr/b3_$cfg.json 2> gpurun_out/lr/b3_$cfg.err
python - <<P
import json
d=json.loads(open("gpurun_out/lr/b3_$cfg.json").read().strip().splitlines()[-1])
print("$cfg", d["value"], d["ms_per_step"])
P
done
