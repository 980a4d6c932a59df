"""A/B timing of kernel variants in one process, interleaved rounds (development aid; run via gpurun).
usage: python tools/ab.py [name=ENV1=val;ENV2=val ...]
env: AB_SCENE=file.ron | AB_SPHERES=n (synthetic plane), AB_W, AB_H, AB_F (frames per launch), AB_ROUNDS;
a variant may set AB_BVH=sah|reference|lbvh (tree builder)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pathtracer_amd as pa
from oracle import orc

variants = {}
for a in sys.argv[1:]:
    name, _, envs = a.partition("=")
    variants[name] = dict(e.split("=", 1) for e in envs.split(";") if e)
if not variants:
    variants = {"lockstep": {"MI355PT_KERNEL": "lockstep"}, "sched": {}}
KEYS = sorted({k for v in variants.values() for k in v})

def bvh_mode():
    return {"sah": pa.BVH_SAH, "reference": pa.BVH_REFERENCE, "lbvh": pa.BVH_LBVH}[os.environ.get("AB_BVH", "sah")]

def setenv(v):
    for k in KEYS:
        os.environ.pop(k, None)
    os.environ.update(v)

import numpy as np
if os.environ.get("AB_SPHERES"):
    from pathtracer_amd import synth
    sc = synth.plane_scene(int(os.environ["AB_SPHERES"]))
    cam = sc.camera
    osc = orc.Scene({"camera": (np.array(cam.pos, np.float32), np.float32(cam.yaw), np.float32(cam.pitch),
                                np.float32(cam.vfov)),
                     "spheres": sc.spheres, "lights": sc.lights, "lambertians": sc.lambertians,
                     "metals": sc.metals, "glass": sc.glass})
else:
    scene_path = os.environ.get("AB_SCENE", "fixtures/scene1.ron")
    sc = pa.load_ron_scene(scene_path)
    osc = orc.Scene.from_ron(scene_path)
# parity first (small)
W, H, seeds = 160, 90, [11, 22, 33]
ref, st = orc.render(osc, osc.camera(W, H), W, H, seeds, threads=16)
for name, v in variants.items():
    setenv(v)
    with pa.Renderer(sc, pa.Settings(1, 50), W, H, bvh_mode=bvh_mode()) as r:
        r.render(seeds=seeds)
        bad = int((r.read_accum().view(np.uint32) != ref.view(np.uint32)).any(axis=2).sum())
        print(f"parity {name}: {'bit-identical' if bad == 0 else f'{bad} of {W*H} pixels differ'} rays {r.stats()['rays']} vs {st['rays']}", flush=True)
        if bad > W * H // int(os.environ.get("AB_PARITY_DIV", 1000)):   # AB_PARITY_DIV=1: time a variant that is known to differ   # the stated bound on branch flips; scenes of the reference's scale are bit-identical
            sys.exit(1)
W, H, F = int(os.environ.get("AB_W", 1920)), int(os.environ.get("AB_H", 1080)), int(os.environ.get("AB_F", 64))
rs = {}
for name, v in variants.items():   # one renderer per variant: scene-build switches apply at creation
    setenv(v)
    rs[name] = pa.Renderer(sc, pa.Settings(1, 50), W, H, bvh_mode=bvh_mode())
res = {n: [] for n in variants}
for rnd in range(int(os.environ.get("AB_ROUNDS", 5))):
    for name, v in variants.items():
        setenv(v)
        r = rs[name]
        r.clear()
        r.render(F)
        s = r.stats()
        res[name].append((s["render_ms"], s["rays"], s["resolve_ms"]))
for name, xs in res.items():
    ms = sorted(x[0] for x in xs)
    rays = xs[0][1]
    acc = sorted(x[2] for x in xs)
    print(f"{name:28s} median {ms[len(ms)//2]:8.3f} ms  min {ms[0]:8.3f} ms  -> {rays/ms[len(ms)//2]/1e3:9.1f} Mrays/s (median)  accumulation {acc[len(acc)//2]:6.3f} ms", flush=True)
