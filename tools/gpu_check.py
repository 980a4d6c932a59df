"""Quick GPU-vs-oracle check and a first timing (development aid, run through gpurun)."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pathtracer_amd as pa
from oracle import orc

def check(name, W, H, seeds, bvh_mode=pa.BVH_SAH, samples=1, depth=50):
    path = f"fixtures/{name}.ron"
    sc = pa.load_ron_scene(path)
    osc = orc.Scene.from_ron(path)
    cam = osc.camera(W, H)
    ref, st = orc.render(osc, cam, W, H, seeds, samples=samples, depth=depth, threads=16)
    r = pa.Renderer(sc, pa.Settings(samples, depth), W, H, bvh_mode=bvh_mode, flags=pa.FLAG_COUNT)
    r.render(seeds=seeds)
    acc = r.read_accum()
    same = np.array_equal(ref.view(np.uint32), acc.view(np.uint32))
    nd = int(np.count_nonzero((ref != acc).any(axis=2)))
    s = r.stats()
    print(f"{name} {W}x{H} frames={len(seeds)} spp={samples} bvh={bvh_mode}: "
          f"{'BIT-IDENTICAL' if same else 'DIFF pixels=%d maxabs=%g' % (nd, np.abs(ref-acc).max())} "
          f"rays gpu={s['rays']} oracle={st['rays']} nodes/ray={s['nodes']/max(1,s['rays']):.2f} "
          f"tests/ray={s['tests']/max(1,s['rays']):.2f} render_ms={s['render_ms']:.3f} lds={s['scene_in_lds']}", flush=True)
    r.close()
    return same

ok = True
for name in ("scene2", "scene3", "scene1"):
    for mode in (pa.BVH_SAH, pa.BVH_REFERENCE):
        ok &= check(name, 160, 90, [11, 22, 33, 44], mode)
ok &= check("scene1", 100, 75, [5, 6], samples=3, depth=7)
print("PARITY", "OK" if ok else "FAILED", flush=True)

# first timing: scene1 1080p
sc = pa.load_ron_scene("fixtures/scene1.ron")
r = pa.Renderer(sc, pa.Settings(1, 50), 1920, 1080)
r.render(4)
for n in (16, 64):
    r.clear()
    t = time.time(); r.render(n); dt = time.time() - t
    s = r.stats()
    print(f"scene1 1080p frames={n}: wall {dt*1e3:.1f} ms, kernel {s['render_ms']:.1f} ms, resolve {s['resolve_ms']:.1f} ms, "
          f"paths {s['paths']/1e6:.1f}M rays {s['rays']/1e6:.1f}M -> {s['paths']/s['render_ms']/1e3:.1f} Mpaths/s "
          f"{s['rays']/s['render_ms']/1e3:.1f} Mrays/s", flush=True)
r.close()
