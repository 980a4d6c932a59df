"""Phase population statistics of the wave-scheduled kernel (development aid; run via gpurun).
usage: python tools/phases.py "24,16,24,16" "1,1,1,1" ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pathtracer_amd as pa
sc = pa.load_ron_scene(os.environ.get("AB_SCENE", "fixtures/scene1.ron"))
W, H, F = 1920, 1080, 8
for th in sys.argv[1:]:
    os.environ["MI355PT_SCHED"] = th
    with pa.Renderer(sc, pa.Settings(1, 50), W, H, flags=pa.FLAG_COUNT) as r:
        r.render(F)
        s = r.stats()
    rays = s["rays"]
    line = f"{th:>14s} {s['render_ms']:7.2f} ms |"
    for k, v in s["phases"].items():
        e, l = v["executions"], v["lanes"]
        line += f" {k}: {e*64/rays:6.2f} slots/ray, {l/max(1,e):5.1f} lanes/exec |"
    print(line, flush=True)
