"""Renders one configuration a few times (profiling target; no torch).  usage: python tools/run_cfg.py N_SPHERES|scene.ron W H FRAMES [REPEATS]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pathtracer_amd as pa
from pathtracer_amd import synth
src, W, H, F = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
rep = int(sys.argv[5]) if len(sys.argv) > 5 else 3
sc = pa.load_ron_scene(src) if src.endswith(".ron") else synth.plane_scene(int(src))
r = pa.Renderer(sc, pa.Settings(1, 50), W, H)
for i in range(rep):
    r.clear()
    r.render(F)
    s = r.stats()
    print(f"{src} {W}x{H}x{F}: kernel {s['render_ms']:.2f} ms, {s['rays']/s['render_ms']/1e3:.1f} Mrays/s, {s['paths']/s['render_ms']/1e3:.1f} Mpaths/s", flush=True)
r.close()
