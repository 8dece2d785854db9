#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02j; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
python tools/ubench_blk16.py 256 256 > $O/ubench.txt 2>&1; head -5 $O/ubench.txt
MICI_AMD_IMPLICIT_KERNEL=blk16 timeout 300 python bench.py --config c4 --steps 5 --warmup 1 --no-cpu-baseline --no-extra-configs > $O/bench_c4.json 2> $O/bench_c4.err
python -c "
import json; d=json.loads(open('$O/bench_c4.json').read().strip().splitlines()[-1]); print('c4 %.4g steps/s' % d['value'], 'kernel ms %.2f' % d['roofline']['kernel_ms_per_launch'], 'frac %.3f' % d['roofline']['frac'])"
bash tools/pmc_hbm.sh c4 implicit_blk16_kernel > $O/pmc_c4.log 2>&1; tail -5 $O/pmc_c4.log; cp gpurun_out/pmc_hbm_c4.json $O/ 2>/dev/null
