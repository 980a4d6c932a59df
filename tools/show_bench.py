"""Prints the main fields of a bench.py JSON line.  usage: python tools/show_bench.py file.json"""
import json, sys
d = json.load(open(sys.argv[1]))
print("value", round(d["value"], 1), d["unit"], "| ms/step", round(d["ms_per_step"], 3), "| kernel ms", round(d["kernel_ms_per_launch"], 3),
      "| resolve ms", round(d["resolve_ms_per_launch"], 3), "| roofline frac", round(d["roofline"]["frac"], 4))
print("valu_issue", d["roofline"].get("valu_issue"))
print("traffic", d["roofline"].get("traffic"), d["roofline"].get("hbm_GBps_from_traffic"))
print("cpu_baseline", d.get("cpu_baseline"))
print("rmse", d.get("rmse_vs_reference_png_8bit"))
