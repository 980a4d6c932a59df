#!/bin/bash
# One rocprofv3 counter pass over a short bench run, path kernel only.  usage: tools/pmc_once.sh <tag> COUNTER...
export TMPDIR=/tmp
TAG=$1; shift
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
timeout -s KILL 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-rmse --no-configs > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    if "sched" in k and "true, true" in k:
        print(k, {n: f"{sum(v)/len(v):.4g}" for n, v in c.items()})
PY
