#!/bin/bash
# c2(iv) sometimes shows a pass time well above its kernel time (a one-off stall inside the timed region, in roughly one
# process out of five).  This records where the time goes: N fresh processes under rocprofv3 --hip-trace --kernel-trace;
# for each, the pass / kernel times, the largest gap between consecutive kernels, and every HIP API call over 2 ms.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
N=${1:-8}
cd /tmp && export TMPDIR=/tmp
for i in $(seq 1 $N); do
  rm -rf /tmp/c2iv_tr
  rocprofv3 --hip-trace --kernel-trace --output-format csv -d /tmp/c2iv_tr -o t -- python $ROOT/bench.py --config c2iv --no-cpu-baseline --no-extra-configs > /tmp/c2iv_tr.log 2>&1
  python - <<PY
import csv, glob, json
line = [l for l in open('/tmp/c2iv_tr.log') if l.startswith('{')]
d = json.loads(line[-1]) if line else None
print('run $i: ms_per_pass %.3f kernel_ms %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms_per_launch']) if d else 'run $i: no result')
k = []
for f in glob.glob('/tmp/c2iv_tr/**/*kernel_trace.csv', recursive=True):
    k += [r for r in csv.DictReader(open(f)) if 'leapfrog_mfma' in r['Kernel_Name']]
k.sort(key=lambda r: int(r['Start_Timestamp']))
gaps = [(int(b['Start_Timestamp']) - int(a['End_Timestamp'])) / 1e6 for a, b in zip(k, k[1:])]
if gaps:
    big = sorted(range(len(gaps)), key=lambda j: -gaps[j])[:3]
    print('   kernels %d, largest gaps between consecutive launches (ms): ' % len(k) + ', '.join('%.2f after #%d' % (gaps[j], j) for j in big))
api = []
for f in glob.glob('/tmp/c2iv_tr/**/*hip_api_trace.csv', recursive=True):
    api += list(csv.DictReader(open(f)))
long_calls = [(r['Function'], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6, int(r['Start_Timestamp'])) for r in api
              if int(r['End_Timestamp']) - int(r['Start_Timestamp']) > 2e6]
t0 = int(k[0]['Start_Timestamp']) if k else 0
for name, ms, st in sorted(long_calls, key=lambda x: x[2]):
    print('   %-28s %8.2f ms  at %+9.2f ms from the first kernel' % (name, ms, (st - t0) / 1e6))
PY
done
