"""The kernels' per-lane code (pt_device.h) and the library's scene packing, run on the CPU by
tests/emul, against the oracle: results must be bit-identical for every tree the library can walk."""
import os

import numpy as np
import pytest

from conftest import FIXTURES, bits_equal, oracle_scene, random_scene

SEEDS = [0x00C0FFEE, 0x12345678, 7]


def test_device_sine_and_hash_match_oracle(emul, orc):
    L = orc.lib()
    for x in np.linspace(-91.3, 91.3, 1001).astype(np.float32):
        assert np.float32(emul.emul_det_sin(float(x))) == np.float32(L.orc_det_sin(float(x)))
    for (w, h) in ((64, 36), (1920, 1080), (3840, 2160), (7, 5)):
        for px, py in ((0, 0), (w - 1, h - 1), (w // 2, h // 2), (w // 3, h - 1)):
            # the oracle adds the frame seed (and maps a zero sum to a constant); use seed 1
            assert (emul.emul_pixel_hash(px, py, w, h) + 1) & 0xFFFFFFFF == L.orc_pixel_seed(1, px, py, w, h)


@pytest.mark.parametrize("name", ["scene1", "scene2", "scene3"])
@pytest.mark.parametrize("topo", [0, 1])
def test_bit_identical_to_oracle(emul, orc, pa, name, topo):
    sc = pa.load_ron_scene(os.path.join(FIXTURES, name + ".ron"))
    osc = oracle_scene(orc, sc)
    W, H = 96, 54
    cam = osc.camera(W, H)
    ref, st = orc.render(osc, cam, W, H, SEEDS, threads=4)
    acc, cnt, info = emul.render(sc, cam, W, H, SEEDS, ref_topo=topo)
    # small scene: the direction-ordered records the kernels keep in LDS; the reference's tree is walked the
    # reference's way from plain 64-byte nodes
    assert info[3] == (3 if topo else 2)
    assert bits_equal(ref, acc)
    assert cnt["rays"] == st["rays"] and cnt["paths"] == st["paths"]
    assert [cnt[k] for k in ("b_light", "b_lambertian", "b_metal", "b_glass", "b_sky")] == \
           [st[k] for k in ("bounce_light", "bounce_lambertian", "bounce_metal", "bounce_glass", "misses")]


@pytest.mark.parametrize("samples,depth", [(1, 0), (1, 1), (2, 3), (4, 50)])
def test_samples_and_depth(emul, orc, pa, samples, depth):
    sc = pa.load_ron_scene(os.path.join(FIXTURES, "scene3.ron"))
    osc = oracle_scene(orc, sc)
    W, H = 48, 27
    cam = osc.camera(W, H)
    ref, _ = orc.render(osc, cam, W, H, SEEDS[:2], samples=samples, depth=depth)
    acc, _, _ = emul.render(sc, cam, W, H, SEEDS[:2], samples=samples, depth=depth)
    assert bits_equal(ref, acc)


@pytest.mark.parametrize("swap_mode", [1, 2])
def test_caller_supplied_tree_with_swapped_children(emul, orc, pa, swap_mode):
    # the reference swaps children at random (bvh.rs:107); any such tree must give the oracle's
    # result for that same tree
    sc = pa.load_ron_scene(os.path.join(FIXTURES, "scene1.ron"))
    nodes = pa.bvh_build_reference(sc.spheres, swap_mode, 99)
    osc = oracle_scene(orc, sc, nodes=orc.bvh_build(sc.spheres, swap_mode, 99))
    W, H = 80, 45
    cam = osc.camera(W, H)
    ref, _ = orc.render(osc, cam, W, H, SEEDS[:2], threads=4)
    acc, _, _ = emul.render(sc, cam, W, H, SEEDS[:2], ref_nodes=nodes)
    assert bits_equal(ref, acc)


def _tie_scene(pa):
    # two coincident spheres with different materials: every hit is an exact tie in t
    sc = pa.Scene.default()
    sp = np.zeros(5, pa.renderer.SPHERE_DT)
    sp[:4] = sc.spheres
    sp[4] = sp[0]               # duplicate of the lambertian at +x ...
    sp[4]["mat_type"] = 0       # ... as a light
    sc.spheres = sp
    return sc


@pytest.mark.parametrize("swap_mode", [0, 1])
@pytest.mark.parametrize("topo", [0, 1])
def test_exact_ties_resolve_like_the_reference_walk(emul, orc, pa, swap_mode, topo):
    # compute.wgsl:129: a later leaf with t == closest.t replaces the record, so the winner is the
    # sphere that comes last in the flattened order, whatever tree the library walks
    sc = _tie_scene(pa)
    W, H = 40, 30
    if swap_mode == 0:
        osc = oracle_scene(orc, sc)
        cam = osc.camera(W, H)
        acc, _, _ = emul.render(sc, cam, W, H, SEEDS, ref_topo=topo)
    else:
        nodes = pa.bvh_build_reference(sc.spheres, 1, 1)
        osc = oracle_scene(orc, sc, nodes=orc.bvh_build(sc.spheres, 1, 1))
        cam = osc.camera(W, H)
        acc, _, _ = emul.render(sc, cam, W, H, SEEDS, ref_nodes=nodes)
    ref, st = orc.render(osc, cam, W, H, SEEDS)
    assert st["bounce_light"] + st["bounce_lambertian"] > 0
    assert bits_equal(ref, acc)


def test_unknown_material_is_black_and_short_circuit_flag(emul, orc, pa):
    sc = pa.load_ron_scene(os.path.join(FIXTURES, "scene2.ron"))
    sc.spheres = sc.spheres.copy()
    sc.spheres["mat_type"][1] = 9  # compute.wgsl:356-358
    osc = oracle_scene(orc, sc)
    W, H = 48, 27
    cam = osc.camera(W, H)
    ref, _ = orc.render(osc, cam, W, H, SEEDS[:1])
    acc, _, _ = emul.render(sc, cam, W, H, SEEDS[:1])
    assert bits_equal(ref, acc)
    # glass short-circuit variant (compute.wgsl:344 read literally).  Total internal reflection needs
    # overlapping glass (a lone sphere always lets a ray out), which scene1 has in a few places.
    sc1 = pa.load_ron_scene(os.path.join(FIXTURES, "scene1.ron"))
    o1 = oracle_scene(orc, sc1)
    W, H = 96, 54
    cam1 = o1.camera(W, H)
    seeds = [1, 2, 3, 4]
    orc.lib().orc_set_glass_always_draws(0)
    try:
        ref1, _ = orc.render(o1, cam1, W, H, seeds, threads=4)
    finally:
        orc.lib().orc_set_glass_always_draws(1)
    acc1, _, _ = emul.render(sc1, cam1, W, H, seeds, glass=0)
    assert bits_equal(ref1, acc1)
    always, _ = orc.render(o1, cam1, W, H, seeds, threads=4)
    assert not bits_equal(always, ref1)


def test_synthetic_plane_scene_crop(emul, orc, pa, monkeypatch):
    from pathtracer_amd import synth
    sc = synth.plane_scene(5000)
    osc = oracle_scene(orc, sc)
    W, H = 320, 180
    cam = osc.camera(W, H)
    crop = (100, 164, 60, 96)
    ref, st = orc.render(osc, cam, W, H, SEEDS[:2], crop=crop, threads=4)
    acc, cnt, info = emul.render(sc, cam, W, H, SEEDS[:2], crop=crop)
    assert info[3] == 4                       # too large for LDS: direction-ordered records in global memory
    assert bits_equal(ref, acc)
    # the SAH tree visits far fewer nodes than the reference's z-only split
    assert cnt["nodes"] * 2 < st["interior_visits"]
    # the same tree through its plain 64-byte nodes: the same boxes, so the same image and the same visits
    monkeypatch.setenv("MI355PT_NODES", "plain")
    acc_p, cnt_p, info_p = emul.render(sc, cam, W, H, SEEDS[:2], crop=crop)
    assert info_p[3] == 0 and bits_equal(ref, acc_p)
    assert cnt_p["nodes"] == cnt["nodes"] and cnt_p["tests"] == cnt["tests"]


def test_sah_tree_is_shallower_in_work_than_reference_tree(emul, orc, pa):
    sc = pa.load_ron_scene(os.path.join(FIXTURES, "scene1.ron"))
    osc = oracle_scene(orc, sc)
    W, H = 64, 36
    cam = osc.camera(W, H)
    _, c_sah, i_sah = emul.render(sc, cam, W, H, SEEDS[:1], ref_topo=0)
    _, c_ref, i_ref = emul.render(sc, cam, W, H, SEEDS[:1], ref_topo=1)
    assert i_ref[0] == 319 and i_ref[1] == 9          # 2n-1 nodes -> n-1 interior, balanced depth
    assert c_sah["nodes"] < c_ref["nodes"]


def test_half_rounding_is_directed(emul):
    import ctypes as C
    rng = np.random.default_rng(5)
    xs = np.concatenate([rng.normal(0, s, 2000) for s in (1e-6, 1e-3, 1.0, 30.0, 3000.0, 1e5)] +
                        [np.array([0.0, -0.0, 65504.0, -65504.0, 65505.0, -65505.0, 1e9, -1e9, 2.0 ** -24, -2.0 ** -24,
                                   2.0 ** -25, 6.1e-5, 1.0, -1.0, 0.333251953125, 1000.5, -2000.0])])
    lo = np.zeros(len(xs), np.float32)
    hi = np.zeros(len(xs), np.float32)
    emul.emul_half_bounds(xs.ctypes.data_as(C.c_void_p), C.c_size_t(len(xs)), lo.ctypes.data_as(C.c_void_p),
                          hi.ctypes.data_as(C.c_void_p))
    assert np.all(lo.astype(np.float64) <= xs) and np.all(xs <= hi.astype(np.float64))
    # tight: both bounds are neighbouring binary16 values (or equal when x is one)
    with np.errstate(over="ignore"):
        as_half = xs.astype(np.float16).astype(np.float64)      # round to nearest
    exact = as_half == xs
    assert np.all(lo[exact] == hi[exact])
    fin = np.isfinite(lo) & np.isfinite(hi) & ~exact
    lo16, hi16 = lo[fin].astype(np.float16), hi[fin].astype(np.float16)
    assert np.all(lo16.astype(np.float32) == lo[fin]) and np.all(hi16.astype(np.float32) == hi[fin])
    assert np.all(np.nextafter(lo16, np.float16(np.inf)) == hi16)
    big = np.abs(xs) > 65504
    assert np.all(np.isinf(hi[big & (xs > 0)])) and np.all(lo[big & (xs > 0)] == 65504)
    assert np.all(np.isinf(lo[big & (xs < 0)])) and np.all(hi[big & (xs < 0)] == -65504)


def test_half_precision_nodes_do_not_change_a_bit(emul, orc, pa, monkeypatch):
    # boxes only cull: conservative binary16 planes must give the very same image
    sc = pa.load_ron_scene(os.path.join(FIXTURES, "scene1.ron"))
    osc = oracle_scene(orc, sc)
    W, H = 96, 54
    cam = osc.camera(W, H)
    ref, st = orc.render(osc, cam, W, H, SEEDS, threads=4)
    monkeypatch.setenv("MI355PT_NODES", "half")
    acc, cnt, info = emul.render(sc, cam, W, H, SEEDS)
    assert info[3] == 1
    assert bits_equal(ref, acc) and cnt["rays"] == st["rays"]
    monkeypatch.setenv("MI355PT_NODES", "float")
    acc_f, cnt_f, info_f = emul.render(sc, cam, W, H, SEEDS)
    assert info_f[3] == 2 and bits_equal(ref, acc_f)   # float nodes, walked as direction-ordered records
    assert info[2] < info_f[2]                       # smaller blob
    assert cnt_f["nodes"] <= cnt["nodes"] <= cnt_f["nodes"] * 1.05   # slightly looser boxes, a few more visits


def test_half_precision_nodes_are_chosen_when_only_they_fit_lds(emul, orc, pa):
    from pathtracer_amd import synth
    sc = synth.plane_scene(800)
    osc = oracle_scene(orc, sc)
    W, H = 160, 90
    cam = osc.camera(W, H)
    crop = (40, 120, 30, 70)
    ref, st = orc.render(osc, cam, W, H, SEEDS[:2], crop=crop, threads=4)
    acc, cnt, info = emul.render(sc, cam, W, H, SEEDS[:2], crop=crop)
    assert info[3] == 1 and info[2] + info[1] * 1024 <= 64 * 1024
    assert bits_equal(ref, acc)
    # a scene with detail far from its median centre keeps float boxes (the planes would round too coarsely)
    far = synth.plane_scene(800)
    far.spheres["pos"][4:, 0] += np.float32(30000.0)
    far.spheres["pos"][4:400, 0] -= np.float32(60000.0)
    _, _, info_far = emul.render(far, cam, 16, 9, SEEDS[:1])
    assert info_far[3] == 4   # float boxes, too large for LDS: ordered records in global memory


@pytest.mark.parametrize("k_top", [1, 64, 4096, 10 ** 9])
def test_top_first_node_order_is_the_same_tree(emul, pa, k_top):
    import ctypes as C
    from pathtracer_amd import synth
    sc = synth.plane_scene(5000)
    rc = emul.emul_check_top_first(C.c_void_p(sc.spheres.ctypes.data), C.c_size_t(len(sc.spheres)), C.c_uint32(min(k_top, 2 ** 32 - 1)))
    assert rc == 0


@pytest.mark.parametrize("seed", range(12))
def test_random_scenes(emul, orc, pa, seed):
    sc = random_scene(pa, 1000 + seed)
    osc = oracle_scene(orc, sc)
    W, H = 40, 24
    cam = osc.camera(W, H)
    samples, depth = (1, 50) if seed % 3 else (2, 6)
    ref, st = orc.render(osc, cam, W, H, SEEDS[:2], samples=samples, depth=depth, threads=4)
    for topo in (0, 1):
        acc, cnt, _ = emul.render(sc, cam, W, H, SEEDS[:2], samples=samples, depth=depth, ref_topo=topo)
        assert bits_equal(ref, acc) and cnt["rays"] == st["rays"]


@pytest.mark.parametrize("seed", range(6))
def test_tree_walk_equals_brute_force(emul, pa, orc, seed):
    # closest hits found by testing every sphere, without any box, give the same image as the tree walk
    sc = random_scene(pa, 3000 + seed)
    osc = oracle_scene(orc, sc)
    W, H = 32, 20
    cam = osc.camera(W, H)
    tree, cnt, _ = emul.render(sc, cam, W, H, SEEDS[:2])
    emul.emul_set_brute_force(1)
    try:
        brute, cnt_b, _ = emul.render(sc, cam, W, H, SEEDS[:2])
    finally:
        emul.emul_set_brute_force(0)
    assert bits_equal(tree, brute) and cnt["rays"] == cnt_b["rays"]
    assert cnt_b["tests"] > cnt["tests"]


@pytest.mark.parametrize("seed", [102866, 104944, 106594, 101557])
def test_reference_walk_follows_the_reference_where_trees_matter(emul, orc, pa, seed):
    # Random scenes (found by tools/soak.py) in which a path travels more than a thousand units inside a huge glass
    # sphere and the sphere test "finds" a root for a ray that misses a small sphere's padded box: which spheres
    # a walk tries then decides a pixel.  In one of them even brute force over all spheres differs from the
    # reference's walk.  The reference-topology mode tries exactly the spheres the reference tries.
    sc = random_scene(pa, seed, duplicates=True)
    osc = oracle_scene(orc, sc)
    W, H = 48, 32
    samples, depth = (1, 50) if seed % 4 else (2, 7)
    cam = osc.camera(W, H)
    ref, st = orc.render(osc, cam, W, H, [seed, 7], samples=samples, depth=depth, threads=4)
    acc, cnt, info = emul.render(sc, cam, W, H, [seed, 7], samples=samples, depth=depth, ref_topo=1)
    assert info[3] == 3
    assert bits_equal(ref, acc) and cnt["rays"] == st["rays"]
    # the same spheres are tried: interior visits differ by construction, sphere tests do not
    assert cnt["tests"] == st["leaf_visits"]


@pytest.mark.parametrize("seed", range(8))
def test_camera_ray_candidate_lists_change_no_bit(emul, pa, orc, seed):
    # camera rays that test their tile's candidate list (tile_list_build, the device's default) find the hits the
    # tree walk finds: same accumulator, same rays, whatever the longest list allowed; with lists off every camera
    # ray walks; a list never holds more candidates than allowed
    sc = random_scene(pa, 7000 + seed)
    osc = oracle_scene(orc, sc)
    W, H = 52, 37   # ragged: tiles cut by the right and the lower edge
    cam = osc.camera(W, H)
    ref, st = orc.render(osc, cam, W, H, SEEDS[:2], threads=4)
    try:
        emul.emul_set_tile_lists(0, 15)
        walk, cnt_w, _ = emul.render(sc, cam, W, H, SEEDS[:2])
        assert emul.tile_list_stats()["list_rays"] == 0
        assert bits_equal(ref, walk) and cnt_w["rays"] == st["rays"]
        served = []
        for max_list in (15, 3, 1, 0):
            emul.emul_set_tile_lists(1, max_list)
            acc, cnt, _ = emul.render(sc, cam, W, H, SEEDS[:2])
            s = emul.tile_list_stats()
            assert bits_equal(ref, acc) and cnt["rays"] == st["rays"], max_list
            assert s["list_tests"] <= s["list_rays"] * max_list
            assert s["list_tiles"] + s["walk_tiles"] == ((W + 7) // 8) * ((H + 7) // 8)
            served.append(s["list_rays"])
        assert served == sorted(served, reverse=True)   # shorter limits serve fewer rays
    finally:
        emul.emul_set_tile_lists(1, 15)


def test_candidate_lists_hold_every_sphere_a_camera_ray_hits(emul, pa, orc):
    # soundness of the cone cull, checked against the oracle's own primary hits: the sphere the reference's walk
    # finds for a camera ray is among the candidates of the ray's tile (scene1: every tile has a list)
    sc = pa.load_ron_scene(os.path.join(FIXTURES, "scene1.ron"))
    osc = orc.Scene.from_ron(os.path.join(FIXTURES, "scene1.ron"))
    W, H = 320, 180
    cam = osc.camera(W, H)
    lists, order = emul.tile_lists(sc, cam, W, H)
    counts = lists[..., 0]
    assert (counts != 0xFFFFFFFF).all() and counts.max() <= 15
    assert (counts == 0).any() and (counts > 1).any()   # sky tiles and tiles with several candidates exist
    # all-light variant of the scene: a pixel then shows the colour of the sphere its camera ray hits first
    n = len(sc.spheres)
    from pathtracer_amd.renderer import VEC16_DT, Scene
    sp = sc.spheres.copy()
    sp["mat_type"], sp["mat_index"] = 0, np.arange(n)
    lights = np.zeros(n, VEC16_DT)
    lights["v"][:, 0] = np.arange(n) + 1.0    # exact in binary32: red channel = sphere index + 1
    probe = Scene(sc.camera, sp, lights, sc.lambertians, sc.metals, sc.glass)
    acc, _ = orc.render(oracle_scene(orc, probe), cam, W, H, [5], threads=4)
    hit = acc[..., 0].astype(np.int64) - 1     # -1: sky (its red is below 1)
    for y in range(H):
        for x in range(W):
            if hit[y, x] >= 0:
                rec = lists[y // 8, x // 8]
                assert hit[y, x] in set(order[rec[1:1 + rec[0]]].tolist()), (x, y)


def test_tie_rank_formula_equals_the_reference_tree(emul, pa):
    # A tree built on the GPU takes its tie ranks (which of two spheres with bit-identical roots wins,
    # compute.wgsl:129) from a sort by z and reference_leaf_rank(n, position) instead of building the reference's
    # tree; the formula must give every sphere the place the reference's own tree (bvh.rs:51-140, no child swap)
    # gives it among its leaves.
    emul.emul_reference_leaf_rank.restype = __import__("ctypes").c_uint32
    rng = np.random.default_rng(5)
    for n in list(range(2, 40)) + [63, 64, 65, 100, 257, 1000]:
        sp = np.zeros(n, pa.renderer.SPHERE_DT)
        sp["pos"] = rng.uniform(-5, 5, (n, 3)).astype(np.float32)
        if n > 6:   # equal keys keep their order (stable sort); -0 equals +0
            sp["pos"][3, 2] = sp["pos"][1, 2]
            sp["pos"][5, 2] = np.float32(-0.0)
            sp["pos"][2, 2] = np.float32(0.0)
            sp["pos"][6, 2] = np.float32(-0.0)
        sp["rad"] = 0.3
        nodes = pa.bvh_build_reference(sp, 0, 1)
        leaves = [int(nd["index"]) for nd in nodes if nd["type"] == 1]
        assert sorted(leaves) == list(range(n))
        by_z = np.argsort(sp["pos"][:, 2], kind="stable") if n >= 4 else np.arange(n)
        got = np.zeros(n, np.int64)
        for pos, sphere in enumerate(by_z):
            got[sphere] = emul.emul_reference_leaf_rank(n, pos)
        want = np.zeros(n, np.int64)
        want[leaves] = np.arange(n)
        assert (got == want).all(), n


def test_sphere_kept_out_of_the_tree(emul, orc, pa, monkeypatch):
    # The sphere far larger than the rest of the scene (the ground of every shipped scene) is in no leaf: every
    # walk tests it first, the candidate lists offer it to every tile.  Same bits with and without; not done for a
    # handful of spheres, nor for a huge sphere that does not touch the rest of the scene.
    W, H = 96, 54
    for name, want in (("scene1", 1), ("scene3", 1), ("scene2", 0)):
        sc = pa.load_ron_scene(os.path.join(FIXTURES, name + ".ron"))
        osc = oracle_scene(orc, sc)
        cam = osc.camera(W, H)
        ref, st = orc.render(osc, cam, W, H, SEEDS)
        acc, cnt, _ = emul.render(sc, cam, W, H, SEEDS)
        assert emul.emul_last_hoisted() == want and bits_equal(ref, acc) and cnt["rays"] == st["rays"]
        if want:
            monkeypatch.setenv("MI355PT_HOIST", "0")
            off, cnt_off, _ = emul.render(sc, cam, W, H, SEEDS)
            monkeypatch.delenv("MI355PT_HOIST")
            assert emul.emul_last_hoisted() == 0 and bits_equal(ref, off)
            assert cnt["nodes"] < cnt_off["nodes"]          # one node visit less per walk, at least
    # the ground doubled (equal roots in every ground hit: the tie rank decides): neither copy is far larger than
    # the rest of the scene, so both stay in the tree
    sc = pa.load_ron_scene(os.path.join(FIXTURES, "scene3.ron"))
    sp = np.zeros(len(sc.spheres) + 1, pa.renderer.SPHERE_DT)
    sp[:-1] = sc.spheres
    ground = int(np.argmax(sc.spheres["rad"]))
    sp[-1] = sc.spheres[ground]
    sp[-1]["mat_type"] = 2 if sc.spheres[ground]["mat_type"] != 2 else 1
    sp[-1]["mat_index"] = 0
    sc.spheres = sp
    osc = oracle_scene(orc, sc)
    cam = osc.camera(W, H)
    ref, _ = orc.render(osc, cam, W, H, SEEDS)
    acc, _, _ = emul.render(sc, cam, W, H, SEEDS)
    assert emul.emul_last_hoisted() == 0 and bits_equal(ref, acc)
    # a huge sphere that the scene does not touch stays in the tree
    sc = pa.load_ron_scene(os.path.join(FIXTURES, "scene1.ron"))
    sp = sc.spheres.copy()
    ground = int(np.argmax(sp["rad"]))
    sp["pos"][ground] = (0.0, -5000.0, 0.0)
    sc.spheres = sp
    osc = oracle_scene(orc, sc)
    cam = osc.camera(W, H)
    ref, _ = orc.render(osc, cam, W, H, SEEDS[:1])
    acc, _, _ = emul.render(sc, cam, W, H, SEEDS[:1])
    assert emul.emul_last_hoisted() == 0 and bits_equal(ref, acc)
