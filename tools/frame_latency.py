import sys, os, time
sys.path.insert(0, os.getcwd())
import pathtracer_amd as pa
sc = pa.load_ron_scene("fixtures/scene1.ron")
for (W, H) in ((1280, 720), (1920, 1080)):
    with pa.Renderer(sc, pa.Settings(1, 50), W, H) as r:
        for _ in range(20): r.render(1)
        t0 = time.perf_counter()
        N = 200
        for _ in range(N): r.render(1)
        dt = (time.perf_counter() - t0) / N
        s = r.stats()
        t0 = time.perf_counter()
        for _ in range(N): r.render(1, wait=False)
        r.wait()
        dq = (time.perf_counter() - t0) / N
        print(W, H, "blocking render(1): %.3f ms per call; queued: %.3f ms; kernel %.3f + accumulation %.3f ms per launch" % (dt * 1e3, dq * 1e3, s["render_ms"] / s["launches"], s["resolve_ms"] / s["launches"]))
