set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r01b
python -m pytest tests -m gpu -q -x > gpurun_out/r01b/pytest_gpu.log 2>&1; tail -3 gpurun_out/r01b/pytest_gpu.log | grep -E "passed|failed|error"
for c in c2 c2i c2iv c2bcss c3 c3b c4 c5; do python bench.py --config $c 2>gpurun_out/r01b/bench_$c.err | tail -1 > gpurun_out/r01b/bench_$c.json; cut -c1-130 gpurun_out/r01b/bench_$c.json; done
cd /tmp && export TMPDIR=/tmp
for c in c2 c3 c4; do rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r01b/prof_$c -o $c -- python $GRAFT_REPO_ROOT/bench.py --config $c --no-cpu-baseline --steps 5 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/r01b/prof_$c.log 2>&1; done
ls -R $GRAFT_REPO_ROOT/gpurun_out/r01b | head -40
