#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02i; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_blk16.py -q > $O/pytest_blk16.log 2>&1; tail -6 $O/pytest_blk16.log
timeout 600 python -m pytest tests/test_gpu_implicit.py tests/test_gpu_full_shards.py -x -q -k "large_kernel_boundaries or diagquad_metric_on_the_team or fixture or c4" > $O/pytest_implicit.log 2>&1; tail -3 $O/pytest_implicit.log
python tools/ubench_blk16.py 256 256 > $O/ubench.txt 2>&1; head -28 $O/ubench.txt | cut -c1-120
for k in blk16; do
  MICI_AMD_IMPLICIT_KERNEL=$k timeout 300 python bench.py --config c4 --steps 5 --warmup 1 --no-cpu-baseline --no-extra-configs > $O/bench_c4_$k.json 2> $O/bench_c4_$k.err
  python -c "
import json; d=json.loads(open('$O/bench_c4_$k.json').read().strip().splitlines()[-1]); print('$k', '%.4g steps/s' % d['value'], 'kernel ms %.2f' % d['roofline']['kernel_ms_per_launch'], 'frac %.3f' % d['roofline']['frac'])"
done
