#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02l; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; grep -E "passed|failed" $O/pytest_gpu.log | tail -2
for c in c3b c4; do
  timeout 300 python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-extra-configs > $O/bench_$c.json 2> $O/bench_$c.err
  python -c "
import json; d=json.loads(open('$O/bench_$c.json').read().strip().splitlines()[-1]); print('$c %.4g steps/s' % d['value'], 'kernel ms %.2f' % d['roofline']['kernel_ms_per_launch'], 'frac %.3f' % d['roofline']['frac'])"
done
