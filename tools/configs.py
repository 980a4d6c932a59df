"""Runs the benchmark configurations on one GPU and prints one JSON line per configuration
(development / documentation aid; run via gpurun).   usage: python tools/configs.py [c1 c2 c3a c3b c4 ...]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pathtracer_amd as pa
from pathtracer_amd import synth

def flops(st):
    b = st["bounces"]
    return st["nodes"] * 48 + st["tests"] * 17 + st["accepted"] * 18 + (b["lambertian"] + b["metal"]) * 60 + b["glass"] * 45 + b["sky"] * 8

def bytes_(st):
    hits = sum(st["bounces"][k] for k in ("light", "lambertian", "metal", "glass"))
    return st["nodes"] * 64 + st["tests"] * 16 + hits * 40

CONFIGS = {
    "c1": ("fixtures/scene1.ron", 400, 225, 16),
    "c2": ("fixtures/scene1.ron", 1920, 1080, 1024),
    "c2_720p": ("fixtures/scene1.ron", 1280, 720, 1000),
    "c3a": ("fixtures/scene2.ron", 1920, 1080, 256),
    "c3b": ("fixtures/scene3.ron", 1920, 1080, 256),
    "c4_100k": (100_000, 3840, 2160, 16),
    "c4": (1_000_000, 3840, 2160, int(os.environ.get("C4_FRAMES", 32))),
    "c5_1gpu": (10_000_000, 3840, 2160, int(os.environ.get("C5_FRAMES", 8))),
}
for name in sys.argv[1:] or ["c1", "c2", "c3a", "c3b"]:
    src, W, H, F = CONFIGS[name]
    t0 = time.time()
    sc = pa.load_ron_scene(src) if isinstance(src, str) else synth.plane_scene(src)
    t_scene = time.time() - t0
    t0 = time.time()
    bvh_mode = {"sah": pa.BVH_SAH, "reference": pa.BVH_REFERENCE, "lbvh": pa.BVH_LBVH}[os.environ.get("CFG_BVH", "sah")]
    r = pa.Renderer(sc, pa.Settings(1, 50), W, H, bvh_mode=bvh_mode)
    t_build = time.time() - t0
    t0 = time.time()
    r.reload_scene(sc)              # a second build: no first-use costs (allocations, code objects)
    t_rebuild = time.time() - t0
    r.render(min(F, 16))           # warm-up (long enough to measure the tile costs once)
    r.clear()
    t0 = time.time()
    r.render(F)
    wall = time.time() - t0
    s = r.stats()
    r.close()
    rc = pa.Renderer(sc, pa.Settings(1, 50), W, H, flags=pa.FLAG_COUNT, bvh_mode=bvh_mode)
    fc = min(F, 4)
    rc.render(fc)
    c = rc.stats()
    rc.close()
    k_ms = s["render_ms"]
    out = {"config": name, "spheres": int(len(sc.spheres)), "size": [W, H], "frames": F,
           "scene_s": round(t_scene, 2), "create_s": round(t_build, 2), "rebuild_s": round(t_rebuild, 3), "wall_ms": round(wall * 1e3, 1),
           "kernel_ms": round(k_ms, 2), "resolve_ms": round(s["resolve_ms"], 2),
           "Mpaths_s": round(s["paths"] / k_ms / 1e3, 1), "Mrays_s": round(s["rays"] / k_ms / 1e3, 1),
           "rays_per_path": round(s["rays"] / s["paths"], 3),
           "nodes_per_ray": round(c["nodes"] / c["rays"], 2), "tests_per_ray": round(c["tests"] / c["rays"], 2),
           "flops_per_ray": round(flops(c) / c["rays"], 1), "bytes_per_ray": round(bytes_(c) / c["rays"], 1),
           "TFLOPs": round(flops(c) / c["rays"] * s["rays"] / k_ms / 1e9, 2),
           "scene_GBps": round(bytes_(c) / c["rays"] * s["rays"] / k_ms / 1e6, 1),
           "bvh_nodes": s["bvh_nodes"], "bvh_depth": s["bvh_depth"], "scene_MB": round(s["scene_bytes"] / 1e6, 2),
           "in_lds": s["scene_in_lds"]}
    print(json.dumps(out), flush=True)
