cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/gap -o g -- python $GRAFT_REPO_ROOT/bench.py --config c2iv --no-cpu-baseline --steps 10 --warmup 2 > /tmp/gap.log 2>&1
python - <<PY
import csv, glob
rows=[]
for f in glob.glob('/tmp/gap/**/*kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
prev=None
for r in rows:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    name=r['Kernel_Name'][:40]
    gap=(s-prev)/1e3 if prev else 0
    print(f"{name:42s} dur {(e-s)/1e3:9.1f} us  gap-from-prev-end {gap:9.1f} us")
    prev=e
PY
tail -2 /tmp/gap.log | cut -c1-200
