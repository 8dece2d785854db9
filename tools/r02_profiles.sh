#!/bin/bash
# Round-2 evidence run (GPU box): default bench line, rocprofv3 kernel stats of the same command, HBM-traffic PMC passes
# for every config, SQ counters for c4 / c3b.  Summaries land in gpurun_out/r02p/ (copied to profiles/ afterwards).
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02p; mkdir -p $O
python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-200 $O/bench_default.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_default -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof_default.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/prof_default -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/rocprofv3_kernel_stats.csv && head -12 $O/rocprofv3_kernel_stats.csv | cut -c1-160
for pair in c2:leapfrog_mfma_kernel c2i:leapfrog_elem c2iv:leapfrog_mfma_kernel c3:implicit_mfma_kernel c3b:softabs_leapfrog_kernel c4:implicit_blk16_kernel c5:constrained_leapfrog_kernel; do
  cfg=${pair%%:*}; kern=${pair##*:}
  bash tools/pmc_hbm.sh $cfg $kern > $O/pmc_hbm_$cfg.log 2>&1; cp gpurun_out/pmc_hbm_$cfg.json $O/ 2>/dev/null
  python -c "
import json; d=json.load(open('$O/pmc_hbm_$cfg.json')); print('$cfg traffic MB/launch %.2f' % (d['traffic_bytes_per_launch']/1e6), len(d['FETCH_SIZE']['per_launch_values_KB']))" 2>&1 | tail -1
done
for pair in c4:implicit_blk16_kernel c3b:softabs_leapfrog_kernel; do
  cfg=${pair%%:*}; kern=${pair##*:}
  bash tools/pmc_collect.sh $cfg $kern > $O/sq_$cfg.log 2>&1; cp gpurun_out/pmc_$cfg/pmc_$cfg.json $O/sq_counters_$cfg.json 2>/dev/null; tail -12 $O/sq_$cfg.log | head -12
done
rm -rf $O/prof_default gpurun_out/pmc_hbm_* gpurun_out/pmc_c4 gpurun_out/pmc_c3b
du -sh $O
