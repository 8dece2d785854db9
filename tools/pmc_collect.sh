#!/bin/bash
# SQ counters of the dominant kernel of a bench config (two PMC passes, own runs: no trace domains mixed in)
#   bash tools/pmc_collect.sh c3 implicit_mfma_kernel ; bash tools/pmc_collect.sh c4 implicit_blk16_kernel
cfg=$1; kern=$2
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$cfg
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU \
  --output-format csv -d $out/p1 -o p1 -- python $GRAFT_REPO_ROOT/bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline --no-extra-configs > $out/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_VMEM \
  --output-format csv -d $out/p2 -o p2 -- python $GRAFT_REPO_ROOT/bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline --no-extra-configs > $out/p2.log 2>&1
python - <<PY
import csv, glob, json, collections
res = collections.defaultdict(list)
for f in glob.glob("$out/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "$kern" in row["Kernel_Name"]:
            res[row["Counter_Name"]].append(float(row["Counter_Value"]))
# one row per (dispatch, counter); average over dispatches
per = {k: sum(v) / max(1, len(v)) for k, v in res.items()}
json.dump({"config": "$cfg", "kernel": "$kern", "dispatches": {k: len(v) for k, v in res.items()}, "counters_per_launch": per},
          open("$out/pmc_$cfg.json", "w"), indent=1)
print(json.dumps(per, indent=1))
PY
