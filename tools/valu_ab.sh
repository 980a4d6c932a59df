#!/bin/bash
# Vector / scalar instruction counts of the bench's path kernel for library builds (one counter pass each).
# usage: tools/valu_ab.sh name1 name2 ...   (name "default" = libmi355pt.so)
export TMPDIR=/tmp
for n in "$@"; do
  if [ "$n" = default ]; then unset MI355PT_LIB; else export MI355PT_LIB=$PWD/pathtracer_amd/libmi355pt_$n.so; fi
  OUT=gpurun_out/valu_$n; rm -rf $OUT; mkdir -p $OUT
  timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/pmc -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-rmse --no-configs > $OUT/log.txt 2>&1 || exit 1
  python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    if "true, true>" in k:
        print("$n", k[-40:], {m: f"{sum(v)/len(v):.5g}" for m, v in c.items()})
PY
done
