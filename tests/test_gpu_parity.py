"""GPU parity: everything goes through the C ABI (ctypes -> libmi355pt.so -> HIP kernels) and is
compared with the CPU oracle on the same seeded inputs.

Bar: BIT-IDENTICAL accumulators.  The arithmetic that reaches a pixel is defined as one IEEE
binary32 operation per source operation on both sides (no contraction, correctly rounded divide and
sqrt), and per-frame results are added in frame order, so no tolerance is needed; the fallback
tolerance the design allows (>= 99.9 % of pixels within 1e-5 relative for one frame) is therefore
never exercised and a single differing bit fails these tests.

One exception, named in its tests: the million- and ten-million-sphere planes.  There, ray-sphere roots of far, small
spheres lose so many bits to cancellation that a root can be "found" for a ray that misses the
sphere's padded box; whether such a root is ever computed depends on which boxes a walk tests, in
the reference as much as here, so a few pixels per ten thousand (all on the horizon) depend on the
tree.  Those tests bound the pixels that differ with the default tree (under 0.1 % at a million spheres, under
1 % at ten million) and require bit-identical results when the reference's own tree is walked the reference's way.
"""
import json
import os

import numpy as np
import pytest

from conftest import FIXTURES, GOLDEN, bits_equal, oracle_scene, random_scene

pytestmark = pytest.mark.gpu

SEEDS = [0x00C0FFEE, 0x12345678, 7, 0xFFFFFFFF]


def gpu_render(pa, sc, W, H, seeds, samples=1, depth=50, **kw):
    with pa.Renderer(sc, pa.Settings(samples, depth), W, H, **kw) as r:
        r.render(seeds=seeds)
        return r.read_accum(), r.stats()


@pytest.mark.parametrize("name", ["scene1", "scene2", "scene3"])
@pytest.mark.parametrize("mode", ["sah", "reference"])
def test_bit_identical_to_oracle(pa, orc, name, mode):
    sc = pa.load_ron_scene(os.path.join(FIXTURES, name + ".ron"))
    osc = oracle_scene(orc, sc)
    W, H = 200, 112
    ref, st = orc.render(osc, osc.camera(W, H), W, H, SEEDS, threads=16)
    acc, gs = gpu_render(pa, sc, W, H, SEEDS, bvh_mode=pa.BVH_REFERENCE if mode == "reference" else pa.BVH_SAH,
                         flags=pa.FLAG_COUNT)
    assert bits_equal(ref, acc)
    assert gs["rays"] == st["rays"] and gs["paths"] == st["paths"]
    b = gs["bounces"]
    assert [b["light"], b["lambertian"], b["metal"], b["glass"], b["sky"]] == \
           [st[k] for k in ("bounce_light", "bounce_lambertian", "bounce_metal", "bounce_glass", "misses")]
    # the default tree is walked from LDS; the reference's tree is walked the reference's way from global memory
    assert gs["scene_in_lds"] == (0 if mode == "reference" else 1)


def test_committed_goldens(pa):
    g = np.load(os.path.join(GOLDEN, "golden.npz"))
    meta = json.load(open(os.path.join(GOLDEN, "golden.json")))
    for name in ("scene1", "scene2", "scene3"):
        sc = pa.load_ron_scene(os.path.join(FIXTURES, name + ".ron"))
        acc, gs = gpu_render(pa, sc, 64, 36, meta["seeds"])
        assert bits_equal(acc, g[name + "_accum"])
        assert gs["rays"] == meta[name + "_stats"]["rays"]
        acc2, _ = gpu_render(pa, sc, 40, 30, meta["seeds"][:2], samples=3, depth=4)
        assert bits_equal(acc2, g[name + "_accum_s3_d4"])
        with pa.Renderer(sc, pa.Settings(), 64, 36) as r:
            assert bits_equal(r.camera_bytes(), g[name + "_camera_16x9"])


@pytest.mark.parametrize("samples,depth", [(1, 0), (1, 1), (2, 3), (5, 50), (1, -1)])
def test_samples_and_depth(pa, orc, samples, depth):
    sc = pa.load_ron_scene(os.path.join(FIXTURES, "scene3.ron"))
    osc = oracle_scene(orc, sc)
    W, H = 96, 54
    ref, st = orc.render(osc, osc.camera(W, H), W, H, SEEDS[:3], samples=samples, depth=depth, threads=8)
    acc, gs = gpu_render(pa, sc, W, H, SEEDS[:3], samples=samples, depth=depth)
    assert bits_equal(ref, acc)
    assert gs["rays"] == st["rays"]


@pytest.mark.parametrize("W,H", [(1, 1), (7, 5), (37, 23), (129, 65), (8, 8), (256, 3)])
def test_ragged_image_sizes(pa, orc, W, H):
    # sizes that do not fill the 8x8 work tiles, down to a single pixel
    sc = pa.load_ron_scene(os.path.join(FIXTURES, "scene2.ron"))
    osc = oracle_scene(orc, sc)
    ref, _ = orc.render(osc, osc.camera(W, H), W, H, SEEDS[:2])
    acc, _ = gpu_render(pa, sc, W, H, SEEDS[:2])
    assert bits_equal(ref, acc)


@pytest.mark.parametrize("swap_mode", [1, 2])
def test_caller_supplied_reference_tree(pa, orc, swap_mode):
    sc = pa.load_ron_scene(os.path.join(FIXTURES, "scene1.ron"))
    nodes = pa.bvh_build_reference(sc.spheres, swap_mode, 4242)
    osc = oracle_scene(orc, sc, nodes=orc.bvh_build(sc.spheres, swap_mode, 4242))
    W, H = 160, 90
    ref, _ = orc.render(osc, osc.camera(W, H), W, H, SEEDS[:2], threads=16)
    acc, gs = gpu_render(pa, sc, W, H, SEEDS[:2], bvh_nodes=nodes)
    assert bits_equal(ref, acc)
    assert gs["bvh_nodes"] == 319


def test_exact_ties_follow_the_reference_order(pa, orc):
    sc = pa.Scene.default()
    sp = np.zeros(5, pa.renderer.SPHERE_DT)
    sp[:4] = sc.spheres
    sp[4] = sp[0]
    sp[4]["mat_type"] = 0
    sc.spheres = sp
    W, H = 64, 48
    osc = oracle_scene(orc, sc)
    ref, st = orc.render(osc, osc.camera(W, H), W, H, SEEDS)
    for mode in (pa.BVH_SAH, pa.BVH_REFERENCE):
        acc, _ = gpu_render(pa, sc, W, H, SEEDS, bvh_mode=mode)
        assert bits_equal(ref, acc)
    nodes = pa.bvh_build_reference(sc.spheres, 1, 1)
    osw = oracle_scene(orc, sc, nodes=orc.bvh_build(sc.spheres, 1, 1))
    ref2, _ = orc.render(osw, osw.camera(W, H), W, H, SEEDS)
    acc2, _ = gpu_render(pa, sc, W, H, SEEDS, bvh_nodes=nodes)
    assert bits_equal(ref2, acc2)
    assert not bits_equal(ref, ref2)  # the swap changes which of the coincident spheres wins


def test_default_scene_unknown_material_and_glass_flag(pa, orc):
    sc = pa.Scene.default()
    osc = oracle_scene(orc, sc)
    W, H = 80, 60
    ref, _ = orc.render(osc, osc.camera(W, H), W, H, SEEDS)
    acc, _ = gpu_render(pa, sc, W, H, SEEDS)
    assert bits_equal(ref, acc)
    sc.spheres = sc.spheres.copy()
    sc.spheres["mat_type"][2] = 77  # compute.wgsl:356-358 -> black
    osc = oracle_scene(orc, sc)
    ref, _ = orc.render(osc, osc.camera(W, H), W, H, SEEDS)
    acc, _ = gpu_render(pa, sc, W, H, SEEDS)
    assert bits_equal(ref, acc)
    sc1 = pa.load_ron_scene(os.path.join(FIXTURES, "scene1.ron"))
    o1 = oracle_scene(orc, sc1)
    orc.lib().orc_set_glass_always_draws(0)
    try:
        ref1, _ = orc.render(o1, o1.camera(96, 54), 96, 54, [1, 2, 3, 4], threads=8)
    finally:
        orc.lib().orc_set_glass_always_draws(1)
    acc1, _ = gpu_render(pa, sc1, 96, 54, [1, 2, 3, 4], flags=pa.FLAG_GLASS_SHORT_CIRCUIT)
    assert bits_equal(ref1, acc1)


def test_renderer_state_machine(pa, orc):
    # mod.rs:132-133,157-166,168-195,242-248
    sc = pa.load_ron_scene(os.path.join(FIXTURES, "scene2.ron"))
    osc = oracle_scene(orc, sc)
    W, H = 64, 36
    with pa.Renderer(sc, pa.Settings(1, 50), W, H, base_seed=99) as r:
        assert r.samples() == -1
        r.render(2)
        assert r.samples() == 2
        sched = [pa.frame_seed(99, f) for f in range(5)]
        ref, _ = orc.render(osc, osc.camera(W, H), W, H, sched[:2])
        assert bits_equal(ref, r.read_accum())
        # reload_settings does not reset the accumulator; the schedule continues at frame 2
        r.reload_settings(pa.Settings(2, 5))
        r.render(1)
        assert r.samples() == 3
        ref, _ = orc.render(osc, osc.camera(W, H), W, H, sched[2:3], samples=2, depth=5, accum=ref)
        assert bits_equal(ref, r.read_accum())
        # a camera edit marks the accumulator dirty: next render clears, frames restart at 1
        cam = pa.CameraSettings((0.5, 1.0, -4.0), 5.0, -3.0, 50.0)
        r.set_camera_config(cam)
        r.reload_settings(pa.Settings(1, 50))
        r.render(1)
        assert r.samples() == 1
        sc2 = pa.load_ron_scene(os.path.join(FIXTURES, "scene2.ron"))
        sc2.camera = cam
        o2 = oracle_scene(orc, sc2)
        ref2, _ = orc.render(o2, o2.camera(W, H), W, H, sched[:1])
        assert bits_equal(ref2, r.read_accum())
        # resize: new aspect, cleared
        r.resize(50, 20)
        r.render(seeds=[5])
        ref3, _ = orc.render(o2, o2.camera(50, 20), 50, 20, [5])
        assert r.read_accum().shape == (20, 50, 4) and bits_equal(ref3, r.read_accum())
        # reload_scene
        sc3 = pa.load_ron_scene(os.path.join(FIXTURES, "scene3.ron"))
        r.reload_scene(sc3)
        r.render(seeds=[6])
        o3 = oracle_scene(orc, sc3)
        ref4, _ = orc.render(o3, o3.camera(50, 20), 50, 20, [6])
        assert bits_equal(ref4, r.read_accum())
        # raw 64-byte camera blob
        r.set_camera_bytes(o3.camera(50, 20))
        r.render(seeds=[6])
        assert bits_equal(ref4, r.read_accum())


def test_frame_order_and_batching_do_not_change_bits(pa, monkeypatch):
    sc = pa.load_ron_scene(os.path.join(FIXTURES, "scene1.ron"))
    W, H = 128, 72
    seeds = list(range(100, 170))  # 70 frames
    with pa.Renderer(sc, pa.Settings(), W, H) as r:
        r.render(seeds=seeds)
        one = r.read_accum()
        assert r.stats()["launches"] == 2             # 16 frames that measure the tile costs, then the other 54
        r.clear()
        r.render(seeds=seeds[:33])
        r.render(seeds=seeds[33:34])
        r.render(seeds=seeds[34:])
        split = r.read_accum()
        r.clear()
        r.render(seeds=seeds)
        again = r.read_accum()
    monkeypatch.setenv("MI355PT_MAX_BATCH", "16")   # development switches are read when a context is created
    with pa.Renderer(sc, pa.Settings(), W, H) as r:
        r.render(seeds=seeds)                       # the same call split into 5 launches
        assert bits_equal(one, r.read_accum()) and r.stats()["launches"] == 5
    monkeypatch.delenv("MI355PT_MAX_BATCH")
    assert bits_equal(one, split) and bits_equal(one, again)
    assert np.all(one[..., 3] == 1.0) and np.isfinite(one).all()


def test_exchange_variant_is_bit_identical(pa, orc, monkeypatch):
    # The path kernel variant whose waves trade paths that wait for a scatter step through two pools in LDS (the
    # default for long launches; forced here).  Which wave scatters a path must not change a bit.
    path = os.path.join(FIXTURES, "scene1.ron")
    sc, osc = pa.load_ron_scene(path), orc.Scene.from_ron(path)
    W, H, seeds = 200, 112, [7, 8, 9, 10, 11]
    ref, st = orc.render(osc, osc.camera(W, H), W, H, seeds, threads=8)
    monkeypatch.setenv("MI355PT_XCHG", "1")
    monkeypatch.setenv("MI355PT_BLOCK", "1024")
    for thr in ("12,8,24,16,6,32", "12,8,24,65,65,32", "20,4,24,40,20,16"):   # defaults; everything thin is parked; odd
        monkeypatch.setenv("MI355PT_XCHG_SCHED", thr)
        with pa.Renderer(sc, pa.Settings(1, 50), W, H, flags=pa.FLAG_COUNT) as r:
            r.render(seeds=seeds)
            s = r.stats()
            assert s["exchange_on"] == 1 and s["exchange"]["lanes_deposited"] == s["exchange"]["lanes_withdrawn"] > 0
            assert bits_equal(ref, r.read_accum()) and s["rays"] == st["rays"]
        with pa.Renderer(sc, pa.Settings(1, 50), W, H) as r:
            r.render(seeds=seeds)
            assert bits_equal(ref, r.read_accum())
    with pa.Renderer(sc, pa.Settings(3, 50), W, H) as r:     # several samples per path: the variant steps aside
        r.render(seeds=seeds[:1])
        assert r.stats()["exchange_on"] == 0


def test_exchange_variant_with_the_tree_in_global_memory(pa, orc, monkeypatch):
    # the same exchange when the scene does not fit LDS (long launches of large scenes run it by default; the
    # 1024-lane kernel is forced here because the test's launch is short)
    from pathtracer_amd import synth
    sc = synth.plane_scene(20000)
    osc = oracle_scene(orc, sc)
    W, H = 320, 180
    crop = (96, 224, 60, 120)
    x0, x1, y0, y1 = crop
    ref, st = orc.render(osc, osc.camera(W, H), W, H, SEEDS[:2], crop=crop, threads=16)
    monkeypatch.setenv("MI355PT_BLOCK", "1024")
    for thr in (None, "12,8,24,65,65,32"):
        if thr:
            monkeypatch.setenv("MI355PT_XCHG_SCHED", thr)
        for flags in (pa.FLAG_COUNT, 0):
            with pa.Renderer(sc, pa.Settings(1, 50), W, H, flags=flags) as r:
                r.render(seeds=SEEDS[:2])
                s = r.stats()
                assert s["scene_in_lds"] == 0 and s["exchange_on"] == 1
                if flags:
                    assert s["exchange"]["lanes_deposited"] == s["exchange"]["lanes_withdrawn"] > 0
                assert bits_equal(ref[y0:y1, x0:x1], r.read_accum()[y0:y1, x0:x1])
    monkeypatch.setenv("MI355PT_XCHG_GLOBAL", "0")
    with pa.Renderer(sc, pa.Settings(1, 50), W, H) as r:
        r.render(seeds=SEEDS[:2])
        assert r.stats()["exchange_on"] == 0
        assert bits_equal(ref[y0:y1, x0:x1], r.read_accum()[y0:y1, x0:x1])


def test_trees_in_global_memory_walked_as_ordered_or_plain_records(pa, orc, monkeypatch):
    # A tree that stays in global memory is walked through direction-ordered 80-byte records built on the device
    # (MI355PT_NODES=plain: through the blob's own 64-byte nodes).  The same boxes either way: the same image, the
    # same node visits and sphere tests, with every workgroup size and with the tree built on the device.
    from pathtracer_amd import synth
    sc = synth.plane_scene(20000)
    osc = oracle_scene(orc, sc)
    W, H = 320, 180
    crop = (96, 224, 60, 120)
    x0, x1, y0, y1 = crop
    ref, st = orc.render(osc, osc.camera(W, H), W, H, SEEDS[:2], crop=crop, threads=16)
    counts = {}
    for nodes in ("ordered", "plain"):
        if nodes == "plain":
            monkeypatch.setenv("MI355PT_NODES", "plain")
        for block in ("512", "1024"):
            monkeypatch.setenv("MI355PT_BLOCK", block)
            for mode in (pa.BVH_SAH, pa.BVH_LBVH):
                with pa.Renderer(sc, pa.Settings(1, 50), W, H, flags=pa.FLAG_COUNT, bvh_mode=mode) as r:
                    r.render(seeds=SEEDS[:2])
                    s = r.stats()
                    assert s["scene_in_lds"] == 0
                    assert bits_equal(ref[y0:y1, x0:x1], r.read_accum()[y0:y1, x0:x1])
                    counts.setdefault((block, mode), []).append((s["nodes"], s["tests"], s["rays"]))
    for k, v in counts.items():
        assert v[0] == v[1], k


def test_queued_renders_equal_blocking_renders(pa, monkeypatch):
    # render_async queues launches without joining; frames must still be added in order, also when a
    # call is split into several launches (forced here: 64 frames per launch)
    monkeypatch.setenv("MI355PT_MAX_BATCH", "64")
    sc = pa.load_ron_scene(os.path.join(FIXTURES, "scene1.ron"))
    W, H = 256, 144
    seeds = np.arange(1, 215, dtype=np.uint32) * np.uint32(2654435761)
    with pa.Renderer(sc, pa.Settings(1, 50), W, H) as r:
        r.render(seeds=seeds)
        blocking, st_b = r.read_accum(), r.stats()
    with pa.Renderer(sc, pa.Settings(1, 50), W, H) as r:
        at = 0
        for n in (70, 3, 64, 13, 64):
            r.render(seeds=seeds[at:at + n], wait=False)
            at += n
        assert at == len(seeds) and r.samples() == len(seeds)
        queued, st_q = r.read_accum(), r.stats()          # read_accum joins
        r.wait()                                          # joining twice is harmless
    assert bits_equal(blocking, queued)
    assert st_q["rays"] == st_b["rays"] and st_q["paths"] == st_b["paths"]
    # 70 frames = 16 that measure the tile costs + 2 x 27; then one launch per call
    assert st_q["launches"] == 7 and st_q["render_ms"] > 0 and st_q["resolve_ms"] > 0


@pytest.mark.parametrize("block", range(4))
def test_random_scenes(pa, orc, block):
    # 10 random scenes per case (overlaps, nesting, duplicates, camera inside spheres, odd glass, unknown
    # materials) x the three trees (equal roots resolve by the reference's leaf order in all three); the tree
    # built on the GPU also gets the same scenes without exact duplicates
    for k in range(10):
        seed = 5000 + 10 * block + k
        W, H = 64, 40
        samples, depth = (1, 50) if k % 3 else (3, 5)
        for dup, modes in ((True, (pa.BVH_SAH, pa.BVH_REFERENCE, pa.BVH_LBVH)), (False, (pa.BVH_LBVH,))):
            sc = random_scene(pa, seed, duplicates=dup)
            osc = oracle_scene(orc, sc)
            ref, st = orc.render(osc, osc.camera(W, H), W, H, SEEDS[:3], samples=samples, depth=depth, threads=16)
            for mode in modes:
                acc, gs = gpu_render(pa, sc, W, H, SEEDS[:3], samples=samples, depth=depth, bvh_mode=mode)
                assert bits_equal(ref, acc), (seed, mode)
                assert gs["rays"] == st["rays"]


@pytest.mark.parametrize("seed", [102866, 104944, 106594, 101557])
def test_reference_walk_follows_the_reference_where_trees_matter(pa, orc, seed):
    # scenes found by tools/soak.py in which the spheres a walk tries decide a pixel (see the CPU test of the
    # same name); the reference-topology mode tries exactly the spheres the reference tries
    sc = random_scene(pa, seed, duplicates=True)
    osc = oracle_scene(orc, sc)
    W, H = 48, 32
    samples, depth = (1, 50) if seed % 4 else (2, 7)
    ref, st = orc.render(osc, osc.camera(W, H), W, H, [seed, 7], samples=samples, depth=depth, threads=16)
    acc, gs = gpu_render(pa, sc, W, H, [seed, 7], samples=samples, depth=depth, bvh_mode=pa.BVH_REFERENCE,
                         flags=pa.FLAG_COUNT)
    assert bits_equal(ref, acc) and gs["rays"] == st["rays"]
    assert gs["tests"] == st["leaf_visits"]


def test_large_scene_in_global_memory(pa, orc):
    # ~20k spheres: the scene blob no longer fits the LDS budget, nodes come from L2/HBM
    from pathtracer_amd import synth
    sc = synth.plane_scene(20000)
    osc = oracle_scene(orc, sc)
    W, H = 320, 180
    crop = (96, 224, 60, 120)
    ref, st = orc.render(osc, osc.camera(W, H), W, H, SEEDS[:2], crop=crop, threads=16)
    acc, gs = gpu_render(pa, sc, W, H, SEEDS[:2], flags=pa.FLAG_COUNT)
    assert gs["scene_in_lds"] == 0
    x0, x1, y0, y1 = crop
    assert bits_equal(ref[y0:y1, x0:x1], acc[y0:y1, x0:x1])


@pytest.mark.parametrize("name", ["scene1", "scene2", "scene3"])
def test_tree_built_on_the_gpu(pa, orc, name):
    # Morton-order tree from lbvh.hip: another tree, the same image
    sc = pa.load_ron_scene(os.path.join(FIXTURES, name + ".ron"))
    osc = oracle_scene(orc, sc)
    W, H = 96, 54
    ref, st = orc.render(osc, osc.camera(W, H), W, H, SEEDS, threads=16)
    acc, gs = gpu_render(pa, sc, W, H, SEEDS, bvh_mode=pa.BVH_LBVH, flags=pa.FLAG_COUNT)
    assert bits_equal(ref, acc) and gs["rays"] == st["rays"]
    assert 1 <= gs["bvh_nodes"] <= len(sc.spheres) - 1 and gs["bvh_depth"] >= 1


def test_tree_built_on_the_gpu_edge_cases(pa, orc):
    # two spheres (one interior node); many spheres at one position (equal Morton codes);
    # the default scene (4 spheres, one of them the 1000-radius ground)
    W, H = 64, 48
    two = pa.Scene.default()
    two.spheres = two.spheres[:2].copy()
    same = pa.Scene.default()
    sp = np.zeros(4 + 40, pa.renderer.SPHERE_DT)
    sp[:4] = same.spheres
    sp[4:] = same.spheres[1]                       # 40 coincident copies of one sphere
    sp[4:]["rad"] = np.linspace(0.05, 0.45, 40).astype(np.float32)   # nested, so no equal roots
    same.spheres = sp
    for sc in (two, same, pa.Scene.default()):
        osc = oracle_scene(orc, sc)
        ref, st = orc.render(osc, osc.camera(W, H), W, H, SEEDS[:3])
        acc, gs = gpu_render(pa, sc, W, H, SEEDS[:3], bvh_mode=pa.BVH_LBVH)
        assert bits_equal(ref, acc) and gs["rays"] == st["rays"]
        assert 1 <= gs["bvh_nodes"] <= len(sc.spheres) - 1   # subtrees of two spheres are folded into leaves


def test_tree_built_on_the_gpu_is_sound_and_breaks_ties_like_the_reference(pa, orc, emul):
    # The packed scene is read back from the device: the tree must be a tree (every sphere in one leaf, every box
    # inside its parent's, every sphere's padded box inside its leaf's), the CPU harness walking that very tree must
    # give the GPU's bits, and where two different spheres have bit-identical roots the winner must be the
    # reference's (compute.wgsl:129: the later leaf of the reference's own tree) -- the tie ranks come from a sort
    # by z on the device, not from the reference's tree.  Scene 50264 of tools/soak.py at 96 x 64 has such a tie
    # between two overlapping glass spheres in texel (47, 27), frame seed 1558188.
    from conftest import device_scene_blob
    sc = random_scene(pa, 50264, duplicates=False)
    osc = oracle_scene(orc, sc)
    W, H, seeds = 96, 64, [1558188, 50264]
    ref, st = orc.render(osc, osc.camera(W, H), W, H, seeds, samples=2, depth=7, threads=16)
    with pa.Renderer(sc, pa.Settings(2, 7), W, H, bvh_mode=pa.BVH_LBVH) as r:
        r.render(seeds=seeds)
        acc = r.read_accum()
        blob, layout = device_scene_blob(r)
    assert bits_equal(ref, acc)
    emul.set_blob(blob, layout)
    try:
        assert emul.emul_check_blob_tree() == 0
        crop = (40, 56, 24, 32)
        x0, x1, y0, y1 = crop
        walked, _, _ = emul.render(sc, osc.camera(W, H), W, H, seeds, samples=2, depth=7, crop=crop)
        assert bits_equal(walked[y0:y1, x0:x1], acc[y0:y1, x0:x1])
    finally:
        emul.set_blob(None)
    for n in (3000, 100_000):   # a larger tree: soundness only
        from pathtracer_amd import synth
        with pa.Renderer(synth.plane_scene(n), pa.Settings(1, 50), 64, 36, bvh_mode=pa.BVH_LBVH) as r:
            blob, layout = device_scene_blob(r)
        emul.set_blob(blob, layout)
        try:
            assert emul.emul_check_blob_tree() == 0
        finally:
            emul.set_blob(None)


@pytest.mark.parametrize("mode", ["sah", "lbvh"])
def test_sphere_kept_out_of_the_tree(pa, orc, monkeypatch, mode):
    # the ground sphere is in no leaf and every walk tests it first (host builder and GPU builder): same bits as with
    # it in the tree and as the oracle, fewer node visits
    bvh = pa.BVH_SAH if mode == "sah" else pa.BVH_LBVH
    path = os.path.join(FIXTURES, "scene1.ron")
    sc, osc = pa.load_ron_scene(path), orc.Scene.from_ron(path)
    W, H = 200, 117
    ref, st = orc.render(osc, osc.camera(W, H), W, H, SEEDS, threads=16)
    acc, gs = gpu_render(pa, sc, W, H, SEEDS, bvh_mode=bvh, flags=pa.FLAG_COUNT)
    monkeypatch.setenv("MI355PT_HOIST", "0")
    off, gs_off = gpu_render(pa, sc, W, H, SEEDS, bvh_mode=bvh, flags=pa.FLAG_COUNT)
    assert bits_equal(ref, acc) and bits_equal(ref, off) and gs["rays"] == gs_off["rays"] == st["rays"]
    assert gs["nodes"] < gs_off["nodes"] and gs["bvh_nodes"] <= gs_off["bvh_nodes"]


def test_tree_built_on_the_gpu_large(pa, orc):
    from pathtracer_amd import synth
    sc = synth.plane_scene(200_000)
    osc = oracle_scene(orc, sc)
    W, H = 320, 180
    crop = (96, 224, 60, 120)
    ref, st = orc.render(osc, osc.camera(W, H), W, H, SEEDS[:2], crop=crop, threads=16)
    with pa.Renderer(sc, pa.Settings(1, 50), W, H, bvh_mode=pa.BVH_LBVH) as r:
        r.render(seeds=SEEDS[:2])
        acc, gs = r.read_accum(), r.stats()
        r.reload_scene(sc)                      # rebuild on the device, same result
        r.render(seeds=SEEDS[:2])
        again = r.read_accum()
    assert gs["scene_in_lds"] == 0 and len(sc.spheres) // 2 <= gs["bvh_nodes"] <= len(sc.spheres) - 1
    x0, x1, y0, y1 = crop
    assert bits_equal(ref[y0:y1, x0:x1], acc[y0:y1, x0:x1]) and bits_equal(acc, again)


def test_half_precision_nodes_in_lds(pa, orc):
    # ~800 spheres fit the LDS budget only with 32-byte nodes; conservative boxes change no bit
    from pathtracer_amd import synth
    sc = synth.plane_scene(800)
    osc = oracle_scene(orc, sc)
    W, H = 160, 90
    ref, st = orc.render(osc, osc.camera(W, H), W, H, SEEDS[:3], threads=16)
    acc, gs = gpu_render(pa, sc, W, H, SEEDS[:3], flags=pa.FLAG_COUNT)
    assert gs["scene_in_lds"] == 1 and gs["node_format"] == 1
    assert bits_equal(ref, acc) and gs["rays"] == st["rays"]


def test_million_spheres_within_the_stated_bound(pa, orc):
    # deep tree (spilling stack, node prefix in LDS, 68 MB blob); see the module docstring for the bound
    from pathtracer_amd import synth
    sc = synth.plane_scene(1_000_000)
    osc = oracle_scene(orc, sc)
    W, H = 160, 90
    seeds = [11, 22, 33]
    ref, st = orc.render(osc, osc.camera(W, H), W, H, seeds, threads=16)
    acc, gs = gpu_render(pa, sc, W, H, seeds)
    assert gs["scene_in_lds"] == 0 and gs["bvh_depth"] > 12
    differ = (acc.view(np.uint32) != ref.view(np.uint32)).any(axis=2)
    assert differ.sum() * 1000 < W * H
    rows = np.unique(np.argwhere(differ)[:, 0])
    assert len(rows) <= 2                         # the horizon
    assert abs(int(gs["rays"]) - int(st["rays"])) <= 8
    # walking the reference's own tree the reference's way leaves no such pixel
    acc_r, gs_r = gpu_render(pa, sc, W, H, seeds, bvh_mode=pa.BVH_REFERENCE)
    assert bits_equal(ref, acc_r) and gs_r["rays"] == st["rays"]


def test_ten_million_spheres_reference_walk(pa, orc):
    # 680 MB blob, a 26-level tree.  With the default tree 0.2 % of the pixels of this scene depend on the tree
    # (far, tiny spheres; see the module docstring); walking the reference's tree the reference's way leaves none.
    from pathtracer_amd import synth
    sc = synth.plane_scene(10_000_000)
    osc = oracle_scene(orc, sc)
    W, H = 96, 54
    seeds = [11, 22]
    ref, st = orc.render(osc, osc.camera(W, H), W, H, seeds, threads=16)
    acc, gs = gpu_render(pa, sc, W, H, seeds)
    differ = (acc.view(np.uint32) != ref.view(np.uint32)).any(axis=2)
    assert differ.sum() * 1000 <= 3 * W * H        # measured: 33 of 14 400 at 160x90 (0.23 %); bound 0.3 %
    acc_r, gs_r = gpu_render(pa, sc, W, H, seeds, bvh_mode=pa.BVH_REFERENCE)
    assert bits_equal(ref, acc_r) and gs_r["rays"] == st["rays"]


def test_watchdog_leaves_the_accumulator_as_it_was(pa, monkeypatch):
    # A launch whose waves stop finishing paths for longer than the watchdog allows ends with an error, and the
    # accumulation kernel adds nothing of it: the accumulator still holds exactly the launches that completed.
    # Provoked with a 1 ms watchdog on the slow fidelity mode (the reference's z-only tree over 300 k spheres:
    # tens of thousands of node visits per ray).
    from pathtracer_amd import synth
    sc = synth.plane_scene(300_000)
    small = pa.load_ron_scene(os.path.join(FIXTURES, "scene2.ron"))
    monkeypatch.setenv("MI355PT_WATCHDOG_MS", "1")
    with pa.Renderer(small, pa.Settings(1, 50), 96, 54) as r:
        r.render(seeds=[3, 4])                       # quick paths: the watchdog stays quiet
        before = r.read_accum()
    with pa.Renderer(sc, pa.Settings(1, 50), 96, 54, bvh_mode=pa.BVH_REFERENCE) as r:
        r.clear()
        cleared = r.read_accum()
        with pytest.raises(pa.Mi355ptError, match="watchdog"):
            r.render(seeds=[1])
        assert bits_equal(cleared, r.read_accum())   # nothing of the abandoned launch was added
        assert r.samples() == 0 and r.stats()["paths"] == 0   # ... and nothing of it is counted
        with pytest.raises(pa.Mi355ptError, match="watchdog"):
            r.render(seeds=[2])                      # reported per launch, not once and for all
        assert bits_equal(cleared, r.read_accum())
        with pytest.raises(pa.Mi355ptError, match="watchdog"):
            r.render(3, wait=False)                  # queued frames are dropped the same way when the queue is joined
            r.wait()
        s = r.stats()
        assert r.samples() == 0 and s["paths"] == 0 and s["launches"] == 0
    assert np.all(before[..., 3] == 1.0)


def test_validation_errors(pa):
    sc = pa.load_ron_scene(os.path.join(FIXTURES, "scene2.ron"))
    bad = pa.load_ron_scene(os.path.join(FIXTURES, "scene2.ron"))
    bad.spheres = bad.spheres[:1]
    with pytest.raises(pa.Mi355ptError):
        pa.Renderer(bad, pa.Settings(), 16, 9)
    bad = pa.load_ron_scene(os.path.join(FIXTURES, "scene2.ron"))
    bad.spheres = bad.spheres.copy()
    bad.spheres["mat_index"][0] = 100
    with pytest.raises(pa.Mi355ptError):
        pa.Renderer(bad, pa.Settings(), 16, 9)
    bad.spheres["mat_index"][0] = 0
    bad.spheres["pos"][1, 0] = np.inf
    with pytest.raises(pa.Mi355ptError):
        pa.Renderer(bad, pa.Settings(), 16, 9)
    with pa.Renderer(sc, pa.Settings(), 16, 9) as r:
        nodes = pa.bvh_build_reference(sc.spheres)
        broken = nodes.copy()
        broken["index"][0] = 3  # escape index that does not span the tree
        with pytest.raises(pa.Mi355ptError):
            r.reload_scene(sc, bvh_nodes=broken)
        with pytest.raises(pa.Mi355ptError):
            r.reload_settings(pa.Settings(0, 50))
        r.reload_scene(sc)
        r.render(1)  # still usable after the rejected calls


def test_full_size_properties_scene1_1080p(pa, orc):
    # the headline configuration's size; the oracle is too slow here, so check what does not depend on it
    sc = pa.load_ron_scene(os.path.join(FIXTURES, "scene1.ron"))
    W, H = 1920, 1080
    seeds = [pa.frame_seed(0x00C0FFEE, f) for f in range(32)]
    with pa.Renderer(sc, pa.Settings(), W, H) as r:
        r.render(seeds=seeds)
        a = r.read_accum()
        s = r.stats()
        r.clear()
        r.render(seeds=seeds[:16])
        r.render(seeds=seeds[16:])
        b = r.read_accum()
    assert bits_equal(a, b)                                   # idempotent / split-invariant
    assert np.all(a[..., 3] == 1.0) and np.isfinite(a).all() and a[..., :3].min() >= 0.0
    assert s["paths"] == W * H * 32 and s["rays"] >= s["paths"]
    # oracle on a crop of the same frames
    osc = oracle_scene(orc, sc)
    crop = (900, 1028, 500, 564)
    ref, _ = orc.render(osc, osc.camera(W, H), W, H, seeds, crop=crop, threads=16)
    x0, x1, y0, y1 = crop
    assert bits_equal(ref[y0:y1, x0:x1], a[y0:y1, x0:x1])


def test_converged_image_matches_reference_png(pa):
    # fixtures/mdimg.png is the reference's own render of scene1 at 1280x720 (offline.rs defaults:
    # 1000 frames x 1 spp, depth 50).  Its seeds are unknown, so parity is statistical: RMSE in
    # gamma space of 8-bit values.  Both images carry Monte-Carlo noise of ~1000 spp.
    from PIL import Image
    ref = np.asarray(Image.open(os.path.join(FIXTURES, "mdimg.png")).convert("RGB")).astype(np.float64)
    sc = pa.load_ron_scene(os.path.join(FIXTURES, "scene1.ron"))
    with pa.Renderer(sc, pa.Settings(1, 50), 1280, 720) as r:
        r.render(1000)
        img = r.to_rgb8().astype(np.float64)
    rmse = np.sqrt(np.mean((img - ref) ** 2))
    print("RMSE vs mdimg.png (8-bit units):", rmse)
    assert rmse <= 3.0


def test_oracle_image_of_the_reference_setup_byte_for_byte(pa):
    # One chain on one input: tests/golden/mdimg_oracle_1000.png is the ORACLE's rgb8 image of the reference's
    # offline set-up (scene1, 1280x720, 1000 frames, default seed schedule; made by make_mdimg_check.py, which
    # also records its RMSE of 0.79 / 255 against the reference's own mdimg.png).  The GPU path, through the C ABI
    # incl. mi355pt_to_rgb8, must give those bytes exactly.
    import json
    from PIL import Image
    gold = np.asarray(Image.open(os.path.join(GOLDEN, "mdimg_oracle_1000.png")).convert("RGB"))
    with open(os.path.join(GOLDEN, "mdimg_check.json")) as f:
        meta = json.load(f)
    sc = pa.load_ron_scene(os.path.join(FIXTURES, "scene1.ron"))
    with pa.Renderer(sc, pa.Settings(1, 50), meta["width"], meta["height"], base_seed=meta["base_seed"]) as r:
        r.render(meta["frames"])
        img = r.to_rgb8()
        assert r.stats()["rays"] == meta["readings"]["always_draws"]["rays"]
    assert img.shape == gold.shape and np.array_equal(img, gold)
    with pa.Renderer(sc, pa.Settings(1, 50), meta["width"], meta["height"], base_seed=meta["base_seed"],
                     flags=pa.FLAG_GLASS_SHORT_CIRCUIT) as r:
        r.render(meta["frames"])
        assert r.stats()["rays"] == meta["readings"]["short_circuit"]["rays"]
        other = r.to_rgb8()
    # the other reading of compute.wgsl:344 is a different image (631 pixels), equally far from mdimg.png
    assert int(np.count_nonzero((other != gold).any(axis=2))) == meta["between_readings"]["pixels_differing"]


def test_accumulator_is_visible_to_torch_without_a_copy(pa):
    # the multi-GPU path hands the library's accumulator to torch.distributed; run it the way
    # bench.py does (torch first, then the library) in a fresh process
    import subprocess
    import sys
    code = r'''
import os, sys
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import pathtracer_amd as pa
from pathtracer_amd import parallel
sc = pa.load_ron_scene("fixtures/scene2.ron")
with pa.Renderer(sc, pa.Settings(), 64, 36) as r:
    r.render(seeds=[1, 2, 3])
    t = parallel.accum_tensor(r, 0)
    host = r.read_accum()
    assert t.shape == (36, 64, 4) and t.is_cuda
    assert np.array_equal(t.cpu().numpy().view(np.uint32), host.view(np.uint32))
    t[..., 3] = 7.0                      # torch writes land in the library's buffer
    torch.cuda.synchronize()
    assert np.all(r.read_accum()[..., 3] == 7.0)
    parallel.reduce_accum(t, 1)
    torch.cuda.synchronize()
    assert np.all(r.read_accum()[..., 3] == 1.0)
print("ok")
'''
    from conftest import ROOT
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout + out.stderr


def test_camera_ray_candidate_lists_are_bit_identical(pa, orc, monkeypatch):
    # Camera rays test their tile's candidate list instead of walking the tree (pt_tile_lists_kernel, rebuilt when
    # camera, scene or image size change).  With lists, without, and with lists so short that most tiles keep the
    # walk, the accumulator equals the oracle's; so it does after a camera move, a resize and a scene reload on one
    # context, with several samples per path, and for a tree in global memory.
    path = os.path.join(FIXTURES, "scene1.ron")
    sc, osc = pa.load_ron_scene(path), orc.Scene.from_ron(path)
    W, H, seeds = 200, 117, [21, 22, 23]
    ref, st = orc.render(osc, osc.camera(W, H), W, H, seeds, threads=8)
    tiles = ((W + 7) // 8) * ((H + 7) // 8)
    for env, want_tiles in (({}, tiles), ({"MI355PT_TILE_LISTS": "0"}, 0), ({"MI355PT_TILE_LIST_MAX": "1"}, None)):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        for flags in (0, pa.FLAG_COUNT):
            with pa.Renderer(sc, pa.Settings(1, 50), W, H, flags=flags) as r:
                r.render(seeds=seeds)
                s = r.stats()
                assert bits_equal(ref, r.read_accum()) and s["rays"] == st["rays"], env
                if want_tiles == 0:
                    assert s["list_tiles"] == 0
                elif want_tiles:                       # a few tiles of this small image see more than 15 spheres
                    assert 0.9 * want_tiles < s["list_tiles"] <= want_tiles
                else:
                    assert 0 < s["list_tiles"] < tiles and s["list_candidates"] <= s["list_tiles"]
        for k in env:
            monkeypatch.delenv(k)
    # one context: camera move, resize, scene reload, samples > 1
    other = pa.load_ron_scene(os.path.join(FIXTURES, "scene3.ron"))
    oother = orc.Scene.from_ron(os.path.join(FIXTURES, "scene3.ron"))
    with pa.Renderer(sc, pa.Settings(1, 50), W, H) as r:
        r.render(seeds=seeds)
        assert bits_equal(ref, r.read_accum())
        cam = sc.camera
        moved = pa.CameraSettings((cam.pos[0] + 1.5, cam.pos[1] + 0.5, cam.pos[2] - 2.0), cam.yaw + 30.0, cam.pitch - 5.0, 35.0)
        r.set_camera_config(moved)
        r.render(seeds=seeds)
        def moved_cam(w, h):
            return orc.camera_build(np.array(moved.pos, np.float32), np.float32(moved.yaw), np.float32(moved.pitch),
                                    np.float32(moved.vfov), np.float32(w) / np.float32(h))
        want, _ = orc.render(osc, moved_cam(W, H), W, H, seeds, threads=8)
        assert bits_equal(want, r.read_accum())
        r.resize(96, 64)
        r.render(seeds=seeds[:2])
        want, _ = orc.render(osc, moved_cam(96, 64), 96, 64, seeds[:2], threads=8)
        assert bits_equal(want, r.read_accum())
        r.reload_scene(other)               # the Python mirror also takes the new scene's camera
        r.reload_settings(pa.Settings(3, 7))
        r.render(seeds=seeds[:2])
        want, _ = orc.render(oother, oother.camera(96, 64), 96, 64, seeds[:2], samples=3, depth=7, threads=8)
        assert bits_equal(want, r.read_accum())
        assert r.stats()["list_tiles"] == 12 * 8   # scene3: ten spheres, every tile has a list


def test_sky_only_tiles_accumulated_per_pixel_are_bit_identical(pa, orc, monkeypatch):
    # Tiles whose candidate list is empty show nothing but sky.  Once the tile order has been measured (a launch of
    # at least 8 frames) they sit at the end of the order, and launches of the plain kind leave them to
    # pt_accumulate_kernel: one lane per pixel adds the frames in order, no result buffer.  Same bits as the oracle,
    # same ray count, whether the split is on or off, over several launches and with queued calls.
    for name, (W, H) in (("scene1.ron", (320, 180)), ("scene2.ron", (203, 117))):
        path = os.path.join(FIXTURES, name)
        sc, osc = pa.load_ron_scene(path), orc.Scene.from_ron(path)
        seeds = np.arange(100, 100 + 27, dtype=np.uint32) * np.uint32(2246822519)
        ref, st = orc.render(osc, osc.camera(W, H), W, H, seeds, threads=16)
        for off in (False, True):
            if off:
                monkeypatch.setenv("MI355PT_SKY_TILES", "0")
            with pa.Renderer(sc, pa.Settings(1, 50), W, H) as r:
                r.render(seeds=seeds[:9])                   # measures the tile order: every tile through the path kernel
                assert r.stats()["sky_tiles"] == 0
                r.render(seeds=seeds[9:12])                 # ordered: sky tiles split off
                split = r.stats()["sky_tiles"]
                assert (split == 0) if off else (0 < split < r.stats()["list_tiles"])
                r.render(seeds=seeds[12:20], wait=False)
                r.render(seeds=seeds[20:], wait=False)
                s = r.stats()
                assert bits_equal(ref, r.read_accum()) and s["rays"] == st["rays"] and s["paths"] == st["paths"], (name, off)
                r.reload_settings(pa.Settings(2, 50))       # two samples per path: the split steps aside
                r.render(seeds=seeds[:2])
                assert r.stats()["sky_tiles"] == 0
            if off:
                monkeypatch.delenv("MI355PT_SKY_TILES")


def test_short_pixel_division_is_verified_and_changes_no_bit(pa, orc, monkeypatch):
    # The camera rays' divisions by width and height run as x * (1/d) plus one correction step when the device has
    # found that sequence equal to the correctly rounded quotient for every possible pixel coordinate (checked per
    # image size at resize; stats report whether it is in use).  With it, without it and in the oracle: same bits.
    sc = pa.load_ron_scene(os.path.join(FIXTURES, "scene3.ron"))
    osc = oracle_scene(orc, sc)
    used = {}
    for W, H in ((1920, 1080), (1280, 720), (400, 225), (37, 23), (1, 1), (129, 65), (1000, 3)):
        seeds = SEEDS[:2] if W * H > 500_000 else SEEDS
        with pa.Renderer(sc, pa.Settings(1, 50), W, H) as r:
            r.render(seeds=seeds)
            a, used[(W, H)] = r.read_accum(), r.stats()["short_div"]
        monkeypatch.setenv("MI355PT_SHORT_DIV", "0")
        with pa.Renderer(sc, pa.Settings(1, 50), W, H) as r:
            r.render(seeds=seeds)
            b = r.read_accum()
            assert r.stats()["short_div"] == 0
        monkeypatch.delenv("MI355PT_SHORT_DIV")
        assert bits_equal(a, b), (W, H)
        if W * H <= 100_000:
            ref, _ = orc.render(osc, osc.camera(W, H), W, H, seeds, threads=8)
            assert bits_equal(ref, a), (W, H)
    print("short division in use:", used)
    assert used[(1, 1)] == 1                      # division by one


def test_results_added_by_the_next_launch_are_bit_identical(pa, orc, monkeypatch):
    # Behind a path kernel of the exchange kind the buffered results wait, and the next launch's path kernel adds
    # them in its prologue (a few of its workgroups, before they take paths); whatever is still waiting when the
    # queue is joined, when a launch follows that has no prologue, or when a launch is about to measure the tile
    # order anew, is added by a kernel of its own.  Several launches per call, queued calls, a re-measurement every
    # fourth launch, a clear and a settings change in between, a tiny image whose launches have fewer workgroups
    # than the prologue would use, all against the oracle; and the same with the deferral off.
    path = os.path.join(FIXTURES, "scene1.ron")
    sc, osc = pa.load_ron_scene(path), orc.Scene.from_ron(path)
    monkeypatch.setenv("MI355PT_BLOCK", "1024")          # the exchange kind also for these short launches
    monkeypatch.setenv("MI355PT_MAX_BATCH", "10")        # several launches per call
    monkeypatch.setenv("MI355PT_MEASURE_EVERY", "4")
    seeds = np.arange(1, 65, dtype=np.uint32) * np.uint32(747796405)
    for W, H in ((200, 112), (16, 9)):
        ref, st = orc.render(osc, osc.camera(W, H), W, H, seeds, threads=16)
        ref3, _ = orc.render(osc, osc.camera(W, H), W, H, seeds[:3], threads=16)
        for defer in ("1", "0"):
            monkeypatch.setenv("MI355PT_DEFER_RESOLVE", defer)
            with pa.Renderer(sc, pa.Settings(1, 50), W, H) as r:
                r.render(seeds=seeds[:20])                 # 10 + 10 frames: the first measures the tile order
                r.render(seeds=seeds[20:27], wait=False)   # 7
                r.render(seeds=seeds[27:50], wait=False)   # 8 + 8 + 7, one of them measures again
                r.render(seeds=seeds[50:])                 # 7 + 7
                s = r.stats()
                assert s["exchange_on"] == 1 and s["launches"] == 8
                assert bits_equal(ref, r.read_accum()) and s["rays"] == st["rays"] and r.samples() == len(seeds), (W, H, defer)
                r.clear()
                r.render(seeds=seeds[:3], wait=False)      # something waits while the settings change
                r.reload_settings(pa.Settings(2, 50))      # two samples per path: a kernel without a prologue follows
                r.render(seeds=seeds[3:4], wait=False)
                r.clear()
                r.render(seeds=seeds[:3])
                assert r.samples() == 3
            with pa.Renderer(sc, pa.Settings(1, 50), W, H) as r:
                r.render(seeds=seeds[:3], wait=False)
                got = r.read_accum()                        # joins: the waiting results are added first
                assert bits_equal(ref3, got), (W, H, defer)
