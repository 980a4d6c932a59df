import sys, os
sys.path.insert(0, os.getcwd())
import pathtracer_amd as pa
sc = pa.load_ron_scene("fixtures/scene1.ron")
with pa.Renderer(sc, pa.Settings(1, 50), 1920, 1080, flags=pa.FLAG_COUNT) as r:
    r.render(16); r.clear(); r.render(int(os.environ.get("PH_F", 64)))
    s = r.stats()
    rays = s["rays"]
    print("rays", rays, "nodes/ray", s["nodes"]/rays, "tests/ray", s["tests"]/rays)
    for k, v in s["phases"].items():
        print(f"{k:10s} wave-steps/ray {v['executions']/rays:.4f}  lanes/step {v['lanes']/max(1,v['executions']):.1f}")
    print("exchange", s["exchange_on"], s["exchange"])
    print("render_ms", s["render_ms"])
