#!/bin/bash
# VALU / SALU instruction counts of the bench kernels (one counter pass); usage: tools/valu_count.sh <tag>
export TMPDIR=/tmp
OUT=gpurun_out/valu_${1:-x}
mkdir -p $OUT
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-rmse > $OUT/log.txt 2>&1 || exit 1
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    if "sched" in k:
        print(k, {n: f"{sum(v)/len(v):.4g}" for n, v in c.items()})
PY
