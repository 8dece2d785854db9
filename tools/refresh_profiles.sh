#!/bin/bash
# ONE script regenerates every measured file the bench line and DESIGN.md cite, from the binaries in the tree
# (VERDICT r02 #5).  Run it on the GPU box; results land in gpurun_out/<round>p/ (gpurun merges that back), then
#     bash tools/refresh_profiles.sh collect
# on the build host copies the summaries into profiles/ with the round prefix.
#     gpurun --timeout 1700 -- 'bash tools/refresh_profiles.sh'
# What it produces (ROUND=r05 by default; SKIP_TESTS=1 leaves the pytest run out, SHORT_FUZZ=1 runs 30-case fuzz seeds):
#   <round>_gpu_tests.txt                 tail of `pytest -m gpu`
#   <round>_bench_default.json            the exact default command, `python bench.py`: its ONE compact stdout line
#   <round>_bench_configs.json            ... and the full per-config record it writes beside it (the sidecar)
#   <round>_rocprofv3_kernel_stats.csv    `rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline`
#   <round>_<cfg>_pmc_hbm.json            FETCH_SIZE / WRITE_SIZE passes (separate runs, gfx950 correction) - c2 c2i c2iv c3 c3b c4 c5
#                                         and the user-source configs c3_user c4_general c3b_dense (kernels mm_rtc_*)
#   <round>_<cfg>_sq_counters.json        SQ counters (two passes) - c2 c3 c3b c4 c5 c3_user c4_general c3b_dense
#   <round>_c4_ubench_blk16.txt, <round>_c3_ubench_mfma.txt   phase clocks of the two dense-Riemannian kernels
#   <round>_fuzz_parity.txt               tools/fuzz_parity.py, three seeds x 80 cases + 60 long SoftAbs cases + constrained + user-source kinds
#   <round>_host_latency.txt              tools/host_latency.py in three fresh processes (single-state Integrator.step, step_batch, system.h)
ROUND=${ROUND:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
if [ "$1" = "collect" ]; then
  src=$ROOT/gpurun_out/${ROUND}p
  for f in $src/*; do
    b=$(basename $f)
    case $b in *.json|*.csv|*.txt) cp $f $ROOT/profiles/${ROUND}_$b;; esac
  done
  ls $ROOT/profiles/${ROUND}_* | wc -l
  exit 0
fi
cd $ROOT
O=$ROOT/gpurun_out/${ROUND}p; [ "$PART" = "2" ] || [ -n "$ONLY" ] || rm -rf $O; mkdir -p $O
KERNELS="c2:leapfrog_mfma_kernel c2i:leapfrog_elem c2i_stream:leapfrog_stream_kernel c2iv:leapfrog_mfma_kernel c3:implicit_mfma_kernel c3b:softabs_leapfrog_kernel c4:implicit_blk16_kernel c5:constrained_leapfrog_kernel c3_user:mm_rtc_riem_step c4_general:mm_rtc_riem_step c4_user_lowrank:mm_rtc_riem_step c3b_dense:mm_rtc_softabs_step c4_d512:implicit_global_kernel c3b_d128:softabs_leapfrog_kernel c3b_d256:softabs_leapfrog_kernel"

NF=80; [ -n "$SHORT_FUZZ" ] && NF=30
# PART=1: tests, counters, bench, kernel trace;  PART=2: phase clocks, fuzz, host latency;  unset: everything
if [ "$PART" = "2" ]; then mkdir -p $O; fi
if [ "$PART" != "2" ]; then
# ONLY="c4_d512 c3b_d128": the counter passes of these configs only (a kernel changed after the full run)
if [ -n "$ONLY" ]; then K2=""; for pair in $KERNELS; do for o in $ONLY; do [ "${pair%%:*}" = "$o" ] && K2="$K2 $pair"; done; done; KERNELS=$K2; fi
if [ -z "$SKIP_TESTS" ]; then
  python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -4 > $O/gpu_tests.txt; cat $O/gpu_tests.txt
fi
for pair in $KERNELS; do
  cfg=${pair%%:*}; kern=${pair##*:}
  bash tools/pmc_hbm.sh $cfg $kern > $O/pmc_hbm_$cfg.log 2>&1
  [ -f gpurun_out/pmc_hbm_$cfg.json ] && mv gpurun_out/pmc_hbm_$cfg.json $O/${cfg}_pmc_hbm.json
  python -c "
import json; d=json.load(open('$O/${cfg}_pmc_hbm.json')); print('$cfg HBM traffic MB/launch %.2f over %d launches' % (d['traffic_bytes_per_launch']/1e6, len(d['FETCH_SIZE']['per_launch_values_KB'])))" 2>&1 | tail -1
  rm -rf gpurun_out/pmc_hbm_$cfg $O/pmc_hbm_$cfg.log
done
# the bench line cites the HBM traffic of THESE passes: put them where bench.py looks (profiles/ of this tree) first
for pair in $KERNELS; do cfg=${pair%%:*}; [ -f $O/${cfg}_pmc_hbm.json ] && cp $O/${cfg}_pmc_hbm.json $ROOT/profiles/${ROUND}_${cfg}_pmc_hbm.json; done
MICI_AMD_BENCH_SIDECAR=$O/bench_configs.json python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-160 $O/bench_default.json
wc -c $O/bench_default.json $O/bench_configs.json

export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -o bench -- python $ROOT/bench.py --no-cpu-baseline > $O/prof_default.log 2>&1
cd $ROOT
f=$(find $O/prof_default -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/rocprofv3_kernel_stats.csv && head -9 $O/rocprofv3_kernel_stats.csv | cut -c1-150
rm -rf $O/prof_default

for pair in $KERNELS; do
  cfg=${pair%%:*}; kern=${pair##*:}
  case $cfg in c2i|c2i_stream|c2iv|c3b_d128|c3b_d256) continue;; esac
  bash tools/pmc_collect.sh $cfg $kern > $O/sq_$cfg.log 2>&1
  [ -f gpurun_out/pmc_$cfg/pmc_$cfg.json ] && mv gpurun_out/pmc_$cfg/pmc_$cfg.json $O/${cfg}_sq_counters.json && echo "$cfg SQ counters ok"
  rm -rf gpurun_out/pmc_$cfg $O/sq_$cfg.log
done

fi  # PART != 2
if [ "$PART" = "1" ]; then du -sh $O; exit 0; fi
python tools/ubench_blk16.py 2>&1 | grep -v "^{" > $O/c4_ubench_blk16.txt; tail -10 $O/c4_ubench_blk16.txt
python tools/ubench_primitives.py > $O/c3_ubench_mfma.txt 2>&1; tail -10 $O/c3_ubench_mfma.txt

for seed in 31 32 33; do
  timeout 1200 python tools/fuzz_parity.py --seed $seed --cases $NF > $O/fuzz_$seed.log 2>&1
  echo "seed $seed rc=$? $(tail -1 $O/fuzz_$seed.log)" >> $O/fuzz_parity.txt
  grep "MISMATCH\|Traceback" -B2 $O/fuzz_$seed.log | head -10 >> $O/fuzz_parity.txt
  rm -f $O/fuzz_$seed.log
done
timeout 1200 python tools/fuzz_parity.py --seed 41 --cases 60 --kinds softabs --long > $O/fuzz_41.log 2>&1
echo "seed 41 (SoftAbs only, four times the steps) rc=$? $(tail -1 $O/fuzz_41.log)" >> $O/fuzz_parity.txt
grep "MISMATCH\|Traceback" -B2 $O/fuzz_41.log | head -10 >> $O/fuzz_parity.txt
rm -f $O/fuzz_41.log
timeout 1200 python tools/fuzz_parity.py --seed 51 --cases 80 --kinds constrained > $O/fuzz_51.log 2>&1
echo "seed 51 (constrained only, D up to 1024) rc=$? $(tail -1 $O/fuzz_51.log)" >> $O/fuzz_parity.txt
grep "MISMATCH\|Traceback\|refused" -B2 $O/fuzz_51.log | head -10 >> $O/fuzz_parity.txt
rm -f $O/fuzz_51.log
# user-source kinds (run-time compiled kernels: a translation unit per kernel family and source text, not per dim)
timeout 1500 python tools/fuzz_parity.py --seed 61 --cases 40 --kinds riemann_user > $O/fuzz_61.log 2>&1
echo "seed 61 (user metric, any D <= 279) rc=$? $(tail -1 $O/fuzz_61.log)" >> $O/fuzz_parity.txt
grep "MISMATCH\|Traceback\|refused" -B2 $O/fuzz_61.log | head -10 >> $O/fuzz_parity.txt
grep "^\[" $O/fuzz_61.log | cut -c1-110 >> $O/fuzz_parity.txt
timeout 1500 python tools/fuzz_parity.py --seed 62 --cases 40 --kinds softabs_user > $O/fuzz_62.log 2>&1
echo "seed 62 (user Hessian, D <= 64) rc=$? $(tail -1 $O/fuzz_62.log)" >> $O/fuzz_parity.txt
grep "MISMATCH\|Traceback\|refused" -B2 $O/fuzz_62.log | head -10 >> $O/fuzz_parity.txt
rm -f $O/fuzz_61.log $O/fuzz_62.log
cat $O/fuzz_parity.txt

# host-call latency of the reference's calling pattern (one Integrator.step(state) per step), three fresh processes
for i in 1 2 3; do python tools/host_latency.py 2>&1 | sed "s/^/run $i: /" >> $O/host_latency.txt; done
grep "single-state" $O/host_latency.txt
du -sh $O
