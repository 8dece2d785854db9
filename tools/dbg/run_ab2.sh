mkdir -p gpurun_out/lr
for cfg in c3 c4 c4_d512; do
  timeout 600 python bench.py --config $cfg --no-cpu-baseline --no-extra-configs > gpurun_out/lr/b2_${cfg}.json 2> gpurun_out/lr/b2_${cfg}.err
  python - <<P
import json
d=json.loads(open("gpurun_out/lr/b2_${cfg}.json").read().strip().splitlines()[-1])
print("${cfg} default", d["value"], d["ms_per_step"])
P
done
MICI_AMD_LIB=mici_amd/lib/ab_slotregs.so timeout 600 python bench.py --config c3 --no-cpu-baseline --no-extra-configs > gpurun_out/lr/b2_c3_slotregs.json 2> gpurun_out/lr/b2_c3_slotregs.err
python - <<P
import json
d=json.loads(open("gpurun_out/lr/b2_c3_slotregs.json").read().strip().splitlines()[-1])
print("c3 slotregs", d["value"], d["ms_per_step"])
P
timeout 2400 python -m pytest tests/test_gpu_implicit.py tests/test_gpu_fuzz_slice.py tests/test_gpu_global_tier.py -q -m gpu -x 2>&1 | grep -v "Warning\|np.tanh" | tail -15 > gpurun_out/lr/tests3.txt
cat gpurun_out/lr/tests3.txt
