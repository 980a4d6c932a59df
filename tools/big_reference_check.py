"""Development check: the synthetic plane at a given size, default tree and reference walk against the oracle.
usage: python tools/big_reference_check.py <spheres>"""
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import pathtracer_amd as pa
from pathtracer_amd import synth
from oracle import orc
from conftest import oracle_scene
orc.build()
n = int(sys.argv[1])
sc = synth.plane_scene(n)
t = time.time(); osc = oracle_scene(orc, sc); print("oracle tree", round(time.time() - t, 1), "s", flush=True)
W, H, seeds = 160, 90, [11, 22, 33]
t = time.time(); ref, st = orc.render(osc, osc.camera(W, H), W, H, seeds, threads=16); print("oracle render", round(time.time() - t, 1), "s", flush=True)
for name, mode in (("default", pa.BVH_SAH), ("reference walk", pa.BVH_REFERENCE)):
    t = time.time()
    with pa.Renderer(sc, pa.Settings(1, 50), W, H, bvh_mode=mode) as r:
        r.render(seeds=seeds)
        acc = r.read_accum(); s = r.stats()
    d = (acc.view(np.uint32) != ref.view(np.uint32)).any(axis=2)
    print(name, "pixels differing", int(d.sum()), "of", W * H, "rays", s["rays"], st["rays"], "kernel ms", round(s["render_ms"], 2), "total s", round(time.time() - t, 1), flush=True)
