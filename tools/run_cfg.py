"""Renders one configuration a few times (profiling target; no torch).  usage: python tools/run_cfg.py N_SPHERES|scene.ron W H FRAMES [REPEATS]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pathtracer_amd as pa
from pathtracer_amd import synth
src, W, H, F = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
rep = int(sys.argv[5]) if len(sys.argv) > 5 else 3
sc = pa.load_ron_scene(src) if src.endswith(".ron") else synth.plane_scene(int(src))
if os.environ.get("RUN_CFG_CAMERA"):   # "x,y,z,yaw,pitch,vfov" (degrees) instead of the scene's own camera
    v = [float(x) for x in os.environ["RUN_CFG_CAMERA"].split(",")]
    sc.camera = pa.CameraSettings(tuple(v[:3]), v[3], v[4], v[5])
if os.environ.get("RUN_CFG_COUNT"):    # visit counts of a few frames first (instrumented kernels)
    with pa.Renderer(sc, pa.Settings(1, 50), W, H, flags=pa.FLAG_COUNT) as rc:
        rc.render(2)
        c = rc.stats()
        print(f"nodes/ray {c['nodes']/c['rays']:.2f} tests/ray {c['tests']/c['rays']:.2f} rays/path {c['rays']/c['paths']:.3f} "
              f"list tiles {c['list_tiles']} bvh nodes {c['bvh_nodes']} depth {c['bvh_depth']} scene MB {c['scene_bytes']/1e6:.0f}", flush=True)
r = pa.Renderer(sc, pa.Settings(1, 50), W, H)
for i in range(rep):
    r.clear()
    r.render(F)
    s = r.stats()
    print(f"{src} {W}x{H}x{F}: kernel {s['render_ms']:.2f} ms, {s['rays']/s['render_ms']/1e3:.1f} Mrays/s, {s['paths']/s['render_ms']/1e3:.1f} Mpaths/s", flush=True)
r.close()
