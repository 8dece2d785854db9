cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_c3b; mkdir -p $out
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $out/p1 -o p1 -- python $GRAFT_REPO_ROOT/bench.py --config c3b --steps 1 --warmup 1 --no-cpu-baseline > $out/p1.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SMEM SQ_INSTS_VMEM --output-format csv -d $out/p2 -o p2 -- python $GRAFT_REPO_ROOT/bench.py --config c3b --steps 1 --warmup 1 --no-cpu-baseline > $out/p2.log 2>&1
python - <<PY
import csv, glob, collections
res = collections.defaultdict(list)
for f in glob.glob("$out/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "softabs_leapfrog" in row["Kernel_Name"]:
            res[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in sorted(res.items()): print(k, sum(v)/len(v), len(v))
PY
tail -3 $out/p2.log
