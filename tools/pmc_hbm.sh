#!/bin/bash
# HBM traffic of the dominant kernel of a bench config from two separate rocprofv3 --pmc passes (no trace
# domains mixed in), corrected as MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE x 2).
#   bash tools/pmc_hbm.sh c3 implicit_mfma_kernel ; bash tools/pmc_hbm.sh c4 implicit_blk16_kernel
cfg=$1; kern=$2
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_hbm_$cfg
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --output-format csv -d $out/$ctr -o $ctr -- python $GRAFT_REPO_ROOT/bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-extra-configs > $out/$ctr.log 2>&1
done
python - <<PY
import csv, glob, json
vals = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    v = []
    for f in glob.glob("$out/%s/**/*counter_collection.csv" % ctr, recursive=True):
        for row in csv.DictReader(open(f)):
            if "$kern" in row["Kernel_Name"] and row["Counter_Name"] == ctr:
                v.append(float(row["Counter_Value"]))
    vals[ctr] = v
import statistics
# median over the launches: one launch in a pass occasionally reads 20x the others (another agent of the node shares the
# counters' TCC view); the per-launch values are kept
fetch = statistics.median(vals["FETCH_SIZE"]) if vals["FETCH_SIZE"] else 0.0
write = statistics.median(vals["WRITE_SIZE"]) if vals["WRITE_SIZE"] else 0.0
res = {"FETCH_SIZE": {"per_launch_values_KB": vals["FETCH_SIZE"], "median_KB": fetch},
       "WRITE_SIZE": {"per_launch_values_KB": vals["WRITE_SIZE"], "median_KB": write},
       "kernel": "$kern ($cfg)",
       "correction": "gfx950: FETCH_SIZE counts 64 B per 128 B request -> doubled (MI355X_MICROARCH.md, HBM); WRITE_SIZE as reported",
       "traffic_bytes_per_launch": (2 * fetch + write) * 1024.0,
       "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- python bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-extra-configs (separate passes)"}
json.dump(res, open("$GRAFT_REPO_ROOT/gpurun_out/pmc_hbm_$cfg.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:600])
PY
