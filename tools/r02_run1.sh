#!/bin/bash
# Round 2, first GPU call: full GPU suite (incl. the new full-shard tests), the new bench line with all configs,
# the loud failure of --gpus 2 on a 1-GPU box, midpoint fixture errors, rocprofv3 kernel stats of the bench.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02a; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-400 $O/bench_default.json; tail -3 $O/bench_default.err
python bench.py --gpus 2 --steps 2 --warmup 0 > $O/bench_gpus2.out 2>&1; echo "gpus2 rc=$?"; tail -3 $O/bench_gpus2.out
python tools/midpoint_errors.py > $O/midpoint_errors.json 2> $O/midpoint_errors.err; python -c "
import json; d=json.load(open('$O/midpoint_errors.json')); print({k: '%.1e' % v['max_scaled_err'] for k, v in d.items()})"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_default -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 5 --warmup 1 > $GRAFT_REPO_ROOT/$O/prof_default.log 2>&1
cd $GRAFT_REPO_ROOT
find $O/prof_default -name "*kernel_stats*" | head; f=$(find $O/prof_default -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-200
du -sh $O
