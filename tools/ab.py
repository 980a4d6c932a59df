"""A/B timing of kernel variants in one process, interleaved rounds (development aid; run via gpurun).
usage: python tools/ab.py [name=ENV1=val;ENV2=val ...]"""
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

def setenv(v):
    for k in KEYS:
        os.environ.pop(k, None)
    os.environ.update(v)

scene_path = os.environ.get("AB_SCENE", "fixtures/scene1.ron")
sc = pa.load_ron_scene(scene_path)
osc = orc.Scene.from_ron(scene_path)
# parity first (small)
W, H, seeds = 160, 90, [11, 22, 33]
ref, st = orc.render(osc, osc.camera(W, H), W, H, seeds, threads=16)
for name, v in variants.items():
    setenv(v)
    with pa.Renderer(sc, pa.Settings(1, 50), W, H) as r:
        r.render(seeds=seeds)
        ok = np.array_equal(r.read_accum().view(np.uint32), ref.view(np.uint32))
        print(f"parity {name}: {'bit-identical' if ok else 'MISMATCH'} rays {r.stats()['rays']} vs {st['rays']}", flush=True)
        if not ok:
            sys.exit(1)
W, H, F = int(os.environ.get("AB_W", 1920)), int(os.environ.get("AB_H", 1080)), int(os.environ.get("AB_F", 64))
rs = {}
for name, v in variants.items():   # one renderer per variant: scene-build switches apply at creation
    setenv(v)
    rs[name] = pa.Renderer(sc, pa.Settings(1, 50), W, H)
res = {n: [] for n in variants}
for rnd in range(int(os.environ.get("AB_ROUNDS", 5))):
    for name, v in variants.items():
        setenv(v)
        r = rs[name]
        r.clear()
        r.render(F)
        s = r.stats()
        res[name].append((s["render_ms"], s["rays"]))
for name, xs in res.items():
    ms = sorted(x[0] for x in xs)
    rays = xs[0][1]
    print(f"{name:28s} median {ms[len(ms)//2]:8.3f} ms  min {ms[0]:8.3f} ms  -> {rays/ms[len(ms)//2]/1e3:9.1f} Mrays/s (median)", flush=True)
