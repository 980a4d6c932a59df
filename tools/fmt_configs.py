"""Condenses the JSON lines of tools/configs.py to one short line each.  usage: python tools/configs.py ... | python tools/fmt_configs.py"""
import sys, json
for l in sys.stdin:
    try:
        d = json.loads(l)
    except Exception:
        print(l.rstrip()); continue
    print(d["config"], "create_s", d.get("create_s"), "rebuild_s", d.get("rebuild_s"), "nodes/ray", d.get("nodes_per_ray"), "kernel_ms", d["kernel_ms"], "resolve_ms", d["resolve_ms"], "Mpaths", d["Mpaths_s"], "Mrays", d["Mrays_s"])
