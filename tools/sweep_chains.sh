#!/bin/bash
# developer A/B: c3 throughput against chains-per-GPU for the three D<=64 dense-Riemannian kernels
run() { python bench.py --config c3 --chains-per-gpu $1 --no-cpu-baseline --steps 5 --warmup 1 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.0f steps/s  %.2f ms/launch" % (d["value"], d["roofline"]["kernel_ms_per_launch"]))'; }
for n in ${CHAINS:-256 1024 2048 4096}; do
  echo "mfma             N=$n: $(run $n)"
  echo "team22x3(4 waves) N=$n: $(MICI_AMD_IMPLICIT_KERNEL=team MICI_AMD_TEAM=22 run $n)"
  echo "team15x5(2 waves) N=$n: $(MICI_AMD_IMPLICIT_KERNEL=team run $n)"
  echo "wave             N=$n: $(MICI_AMD_IMPLICIT_KERNEL=wave run $n)"
done
