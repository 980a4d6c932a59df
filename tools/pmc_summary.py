"""Collapses rocprofv3 counter CSVs under gpurun_out/prof_<tag>/ into profiles/<name>/pmc_summary.json and
copies kernel_stats.csv.  usage: python tools/pmc_summary.py <tag> <name>"""
import collections, csv, glob, json, os, shutil, sys
tag, name = sys.argv[1], sys.argv[2]
src = f"gpurun_out/prof_{tag}"
dst = f"profiles/{name}"
os.makedirs(dst, exist_ok=True)
for f in glob.glob(f"{src}/trace/*/*_kernel_stats.csv"):
    shutil.copy(f, f"{dst}/kernel_stats.csv")
out = {}
for f in glob.glob(f"{src}/pmc_*/*/*_counter_collection.csv"):
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for r in rows:
        k = r["Kernel_Name"].split("(")[0]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta[k] = {x: r[x] for x in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "SGPR_Count", "Scratch_Size")}
    for k, c in agg.items():
        o = out.setdefault(k, {"dispatch": meta[k], "counters_avg_per_dispatch": {}})
        for n, v in c.items():
            o["counters_avg_per_dispatch"][n] = {"avg": sum(v) / len(v), "dispatches": len(v)}
json.dump(out, open(f"{dst}/pmc_summary.json", "w"), indent=1)
for k, v in out.items():
    if "render" in k or "resolve" in k:
        print(k, {n: round(c["avg"]) for n, c in v["counters_avg_per_dispatch"].items()})
