"""The stated configurations at their stated sizes (C1-C4), the offline CLI and the several-device verbs, on the GPU
through the C ABI.  Where the oracle covers the whole input the accumulator bits must be equal; at full sizes the
checks are properties that do not depend on it (split-invariance, counts, alpha, finiteness) plus an oracle crop."""
import os
import subprocess

import numpy as np
import pytest

from conftest import FIXTURES, ROOT, bits_equal, oracle_scene

pytestmark = pytest.mark.gpu
CLI = os.path.join(ROOT, "pathtracer_amd", "mi355pt_offline")


def seeds_of(pa, n, first=0, base=0x00C0FFEE):
    return np.array([pa.frame_seed(base, first + k) for k in range(n)], np.uint32)


def test_c1_scene1_400x225x16_whole_image(pa, orc):
    path = os.path.join(FIXTURES, "scene1.ron")
    sc, osc = pa.load_ron_scene(path), orc.Scene.from_ron(path)
    W, H, seeds = 400, 225, seeds_of(pa, 16)
    ref, st = orc.render(osc, osc.camera(W, H), W, H, seeds, threads=16)
    for mode in (pa.BVH_SAH, pa.BVH_REFERENCE, pa.BVH_LBVH):
        with pa.Renderer(sc, pa.Settings(1, 50), W, H, bvh_mode=mode) as r:
            r.render(16)                                    # the library's own seed schedule
            assert bits_equal(ref, r.read_accum()) and r.stats()["rays"] == st["rays"] and r.samples() == 16


def test_c2_scene1_1080p_1024_frames_whole_image(pa, orc):
    # The headline configuration at its stated size (offline.rs:15-16,34: 1920x1080 here, 1024 frames of one sample).
    # The oracle needs minutes for it, the reference-walk mode -- bit-equal to the oracle wherever both have been
    # run, and exactly the reference's sphere tests -- a second: every texel of the default tree's image must equal
    # the reference walk's, the ray counts too, and a 64x32 crop over the three large spheres equals the oracle.
    path = os.path.join(FIXTURES, "scene1.ron")
    sc, osc = pa.load_ron_scene(path), orc.Scene.from_ron(path)
    W, H, F = 1920, 1080, 1024
    out = {}
    for mode in (pa.BVH_SAH, pa.BVH_REFERENCE):
        with pa.Renderer(sc, pa.Settings(1, 50), W, H, bvh_mode=mode) as r:
            r.render(F)
            out[mode] = (r.read_accum(), r.stats())
            assert r.samples() == F
    a, sa = out[pa.BVH_SAH]
    b, sb = out[pa.BVH_REFERENCE]
    assert bits_equal(a, b)
    assert sa["rays"] == sb["rays"] and sa["paths"] == sb["paths"] == W * H * F
    x0, x1, y0, y1 = crop = (928, 992, 560, 592)
    ref, _ = orc.render(osc, osc.camera(W, H), W, H, seeds_of(pa, F), crop=crop, threads=16)
    assert bits_equal(ref[y0:y1, x0:x1], a[y0:y1, x0:x1])
    assert np.unique(a[y0:y1, x0:x1, 0]).size > 300           # the crop is a real image


def full_size_properties(pa, orc, sc, W, H, frames, crop, exact=True):
    seeds = seeds_of(pa, frames)
    with pa.Renderer(sc, pa.Settings(1, 50), W, H) as r:
        r.render(seeds=seeds)
        a, s = r.read_accum(), r.stats()
        r.clear()
        cut = frames // 3
        r.render(seeds=seeds[:cut])
        r.render(seeds=seeds[cut:], wait=False)
        b = r.read_accum()
    assert bits_equal(a, b)                                   # split- and queue-invariant
    assert np.all(a[..., 3] == 1.0) and np.isfinite(a).all() and a[..., :3].min() >= 0.0
    assert s["paths"] == W * H * frames and s["paths"] <= s["rays"] <= 51 * s["paths"]
    osc = oracle_scene(orc, sc)
    x0, x1, y0, y1 = crop
    ref, _ = orc.render(osc, osc.camera(W, H), W, H, seeds, crop=crop, threads=16)
    if exact:
        assert bits_equal(ref[y0:y1, x0:x1], a[y0:y1, x0:x1])
    else:
        differ = (ref[y0:y1, x0:x1].view(np.uint32) != a[y0:y1, x0:x1].view(np.uint32)).any(axis=2)
        assert differ.sum() * 1000 <= differ.size
    s["accum"], s["oracle_scene"], s["seeds"] = a, osc, seeds
    return s


@pytest.mark.parametrize("name", ["scene2.ron", "scene3.ron"])
def test_c3_small_scenes_1080p_256_frames(pa, orc, name):
    sc = pa.load_ron_scene(os.path.join(FIXTURES, name))
    s = full_size_properties(pa, orc, sc, 1920, 1080, 256, (928, 992, 520, 552))
    assert s["scene_in_lds"] == 1


def test_c4_million_spheres_4k_256_frames(pa, orc):
    # The configuration at its stated size: 3840x2160, all 256 frames (four launches), twice (split- and
    # queue-invariance).  Oracle crop 1: 128x64 texels below the horizon, all 256 frames -- equal bits.  Oracle crop
    # 2: 64x96 texels across the horizon (rows 600-695; the plane's horizon is near row 650), where the sphere
    # test's cancellation lets a far, tiny sphere be "hit" by a ray that misses its padded box, so that the spheres a
    # walk tries decide a sample, in the reference as much as here: after 256 frames a texel differs when any of its
    # 256 samples did, so the bound is per texel of the crop (<= 1 %, measured 1 of 1536) and on the size of the
    # difference (one sample of 256).  The reference-walk mode, which is bit-equal there too, takes minutes per 4K
    # frame at this size and has no way to render a crop: test_million_spheres_within_the_stated_bound checks it
    # on the whole image at 160x90.
    from pathtracer_amd import synth
    sc = synth.plane_scene(1_000_000)
    W, H = 3840, 2160
    s = full_size_properties(pa, orc, sc, W, H, 256, (1900, 2028, 1800, 1864), exact=True)
    assert s["scene_in_lds"] == 0 and s["launches"] >= 4
    assert s["list_tiles"] > 0.9 * (W // 8) * (H // 8)        # camera rays of all but the horizon tiles test lists
    x0, x1, y0, y1 = crop = (1900, 1964, 600, 696)
    ref, _ = orc.render(s["oracle_scene"], s["oracle_scene"].camera(W, H), W, H, s["seeds"], crop=crop, threads=16)
    A, R = s["accum"][y0:y1, x0:x1], ref[y0:y1, x0:x1]
    differ = (A.view(np.uint32) != R.view(np.uint32)).any(axis=2)
    assert differ.sum() * 100 <= differ.size
    rel = np.abs(A[..., :3] - R[..., :3]) / np.maximum(1e-3, np.abs(R[..., :3]))
    assert rel.max() <= 2e-2
    assert (R[..., :3] > 0).all() and np.unique(R[..., 2]).size > differ.size // 2   # the crop is a real image


def run_cli(*args, cwd):
    return subprocess.run([CLI, *args], cwd=cwd, capture_output=True, text=True, timeout=300)


def test_offline_cli(pa, tmp_path):
    from PIL import Image
    scene, settings = os.path.join(FIXTURES, "scene1.ron"), os.path.join(FIXTURES, "settings.ron")
    W, H, F = 320, 180, 24
    p = run_cli("--scene", scene, "--settings", settings, "--width", str(W), "--height", str(H), "--frames", str(F),
                "--out", "a.png", "--chunk", "7", cwd=tmp_path)
    assert p.returncode == 0, p.stderr
    lines = p.stdout.strip().splitlines()                     # offline.rs:75-83
    assert lines[0] == f"Img size: {W}, {H}" and lines[1] == f"Samples: {F}" and lines[2].startswith("Time took: ")
    assert lines[2].endswith("s") and f"{F}/{F} | 100.0% " in p.stderr      # offline.rs:54-60
    sc = pa.load_ron_scene(scene)
    with pa.Renderer(sc, pa.load_ron_settings(settings), W, H) as r:
        r.render(F)
        want = r.to_rgb8()
    got = np.asarray(Image.open(tmp_path / "a.png").convert("RGB"))
    assert np.array_equal(got, want)                          # SaveInfo::finish's bytes
    # the frames split over three contexts (here on one device): the same image up to the order of the additions
    p = run_cli("--scene", scene, "--settings", settings, "--width", str(W), "--height", str(H), "--frames", str(F),
                "--out", "b.png", "--devices", "0,0,0", cwd=tmp_path)
    assert p.returncode == 0, p.stderr
    assert p.stdout.splitlines()[1] == f"Samples: {F}"
    split = np.asarray(Image.open(tmp_path / "b.png").convert("RGB")).astype(int)
    assert np.abs(split - want.astype(int)).max() <= 1 and np.count_nonzero(split != want) <= split.size // 1000
    # offline.rs:37: a scene that cannot be loaded falls back to Scene::default() (scene.rs:15-31)
    p = run_cli("--scene", "no_such_scene.ron", "--settings", settings, "--width", "64", "--height", "36", "--frames", "2",
                "--out", "d.png", cwd=tmp_path)
    assert p.returncode == 0 and "default scene" in p.stderr
    from pathtracer_amd.renderer import Scene
    with pa.Renderer(Scene.default(), pa.load_ron_settings(settings), 64, 36) as r:
        r.render(2)
        assert np.array_equal(np.asarray(Image.open(tmp_path / "d.png").convert("RGB")), r.to_rgb8())
    # offline.rs:39: missing settings end the run (unwrap() there, exit status 1 here), nothing is written
    p = run_cli("--scene", scene, "--settings", "no_such_settings.ron", "--out", "e.png", cwd=tmp_path)
    assert p.returncode == 1 and "error" in p.stderr and not (tmp_path / "e.png").exists()
    # with no flags the reference's file names are looked up in the working directory
    p = run_cli("--frames", "1", "--width", "32", "--height", "18", cwd=FIXTURES)
    assert p.returncode == 0 and os.path.exists(os.path.join(FIXTURES, "img.png"))
    os.remove(os.path.join(FIXTURES, "img.png"))


def test_reduce_over_contexts(pa):
    # mi355pt_reduce with every context on this box's one device: the same-device path of the verb; the frame
    # split is mi355pt_partition_frames.  Against one context rendering all frames only the order of the float
    # additions differs (bound: 1e-6 relative).
    sc = pa.load_ron_scene(os.path.join(FIXTURES, "scene1.ron"))
    W, H, F, N = 256, 144, 50, 3
    with pa.Renderer(sc, pa.Settings(1, 50), W, H) as r:
        r.render(F)
        one, st_one = r.read_accum(), r.stats()
    rs = [pa.Renderer(sc, pa.Settings(1, 50), W, H) for _ in range(N)]
    try:
        for k, r in enumerate(rs):
            first, count = pa.partition_frames(F, N, k)
            r.render(seeds=seeds_of(pa, count, first), wait=False)
        pa.reduce(rs, root=1)
        acc, st = rs[1].read_accum(), rs[1].stats()
        assert rs[1].samples() == F and st["paths"] == st_one["paths"] and st["rays"] == st_one["rays"]
        assert np.all(acc[..., 3] == 1.0)
        rel = np.abs(acc[..., :3] - one[..., :3]) / np.maximum(1.0, np.abs(one[..., :3]))
        assert rel.max() <= 2e-6 and np.sqrt(np.mean(rel.astype(np.float64) ** 2)) <= 1e-6
        assert np.array_equal(rs[1].to_rgb8(), pa.postprocess_rgb8(acc, F))
        rs[0].render(seeds=[1])                               # a spent context starts over
        assert rs[0].samples() == 1
        with pytest.raises(pa.Mi355ptError):
            pa.reduce([rs[1], rs[1]], root=0)
        # a group of one is the identity
        before = rs[1].read_accum()
        pa.reduce([rs[1]], root=0)
        assert bits_equal(before, rs[1].read_accum()) and rs[1].samples() == F
    finally:
        for r in rs:
            r.close()


_scene_cache = {}


def plane(n):
    from pathtracer_amd import synth
    if n not in _scene_cache:
        _scene_cache[n] = synth.plane_scene(n)
    return _scene_cache[n]


def test_c5_one_rank_share_of_16384_frames(pa):
    # C5 at its stated length, as one of its eight ranks sees it: the 10 M-sphere scene at 3840x2160 and the 2 048
    # frames mi355pt_partition_frames gives rank 3 of 8 out of 16 384 (frames 6 144 .. 8 191 of the schedule) -- 32
    # launches of 64 frames, so the queue, the seed ring, the two result buffers and the refresh of the tile order
    # run as long as they do in the configuration.  Once as one blocking call, once as four queued calls: same bits,
    # same counts.  (What no one-GPU box can add is the reduce between the eight ranks.)
    W, H, total, ranks, rank = 3840, 2160, 16384, 8, 3
    first, count = pa.partition_frames(total, ranks, rank)
    assert (first, count) == (6144, 2048)
    seeds = seeds_of(pa, count, first)
    with pa.Renderer(plane(10_000_000), pa.Settings(1, 50), W, H, bvh_mode=pa.BVH_LBVH) as r:
        r.render(seeds=seeds)
        a, sa = r.read_accum(), r.stats()
        assert r.samples() == count
        r.clear()
        for k in range(4):
            r.render(seeds=seeds[k * 512:(k + 1) * 512], wait=False)
        b, sb = r.read_accum(), r.stats()
    assert bits_equal(a, b)
    assert sa["launches"] >= 32 and sa["paths"] == sb["paths"] == W * H * count and sa["rays"] == sb["rays"]
    assert np.all(a[..., 3] == 1.0) and np.isfinite(a).all() and a[..., :3].min() >= 0.0
    assert a[..., :3].max() <= 1.0001 * count       # the generator's one light is (1, 1, 1), the sky at most that


def test_c5_ten_million_spheres_partitioned_over_eight_contexts(pa):
    # C5 in miniature: the 10 M-sphere scene at 4K, its frames split eight ways by mi355pt_partition_frames
    # (the eight contexts share this box's one device; on an eight-GPU node each would have its own and the sum
    # would go through RCCL), one mi355pt_reduce to the root.  256 frames, 32 per context: more than one launch in
    # the one-context render, and every context measures its tile order inside the partitioned path.  Against one
    # context that renders every frame only the order of the 256 float additions per texel differs (root mean
    # square 1e-6 relative; a single texel at most 2e-5, i.e. 256 roundings one way).  Trees built on the device:
    # eight host builds of this scene would take a quarter of a minute.
    sc = plane(10_000_000)
    W, H, F, N = 3840, 2160, 256, 8      # 32 frames per context: the tile order is measured inside the partitioned path
    with pa.Renderer(sc, pa.Settings(1, 50), W, H, bvh_mode=pa.BVH_LBVH) as r:
        r.render(F)
        one, st_one = r.read_accum(), r.stats()
    assert st_one["scene_in_lds"] == 0 and st_one["paths"] == W * H * F and st_one["launches"] > 1
    rs = [pa.Renderer(sc, pa.Settings(1, 50), W, H, bvh_mode=pa.BVH_LBVH) for _ in range(N)]
    try:
        for k, r in enumerate(rs):
            first, count = pa.partition_frames(F, N, k)
            assert count == F // N
            r.render(seeds=seeds_of(pa, count, first), wait=False)
        pa.reduce(rs, root=0)
        acc, st = rs[0].read_accum(), rs[0].stats()
        assert rs[0].samples() == F and st["paths"] == st_one["paths"] and st["rays"] == st_one["rays"]
        assert np.all(acc[..., 3] == 1.0) and np.isfinite(acc).all()
        rel = np.abs(acc[..., :3] - one[..., :3]) / np.maximum(1.0, np.abs(one[..., :3]))
        assert rel.max() <= 2e-5 and np.sqrt(np.mean(rel.astype(np.float64) ** 2)) <= 1e-6
    finally:
        for r in rs:
            r.close()


def test_reduce_comm_with_a_one_rank_rccl_communicator(pa):
    # mi355pt_reduce_comm with a communicator the caller made (here: RCCL's ncclCommInitAll over this box's one
    # device, through ctypes): the ncclReduce the library issues on the context's stream, on a group of one, must
    # leave the sums as they are, put alpha back to 1 and record the frame total.  What a box with one GPU can
    # exercise of the RCCL path; a reduce between two physical GPUs needs a larger machine.
    import ctypes as C
    try:
        rccl = C.CDLL("librccl.so.1")
    except OSError:
        rccl = C.CDLL("/opt/rocm/lib/librccl.so.1")
    comm = C.c_void_p()
    devs = (C.c_int * 1)(0)
    assert rccl.ncclCommInitAll(C.byref(comm), 1, devs) == 0
    try:
        sc = pa.load_ron_scene(os.path.join(FIXTURES, "scene3.ron"))
        with pa.Renderer(sc, pa.Settings(1, 50), 128, 72) as r:
            r.render(5)
            before = r.read_accum()
            rc = pa.lib().mi355pt_reduce_comm(r._ctx, comm, 0, 0, 5)
            assert rc == 0, pa.lib().mi355pt_last_error(r._ctx)
            assert bits_equal(before, r.read_accum()) and r.samples() == 5
            r.render(2)                                        # the root keeps accumulating
            assert r.samples() == 7
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)


def test_reduce_comm_between_two_processes(pa, tmp_path):
    # The one-process-per-GPU shape with the library's own RCCL call: two processes, a two-rank communicator
    # (ncclGetUniqueId / ncclCommInitRank), each renders its half of ten frames and calls mi355pt_reduce_comm.  This
    # box has one GPU, so both ranks sit on device 0: if RCCL accepts that, the root's accumulator must equal a
    # one-process render up to the order of two additions; if it refuses (NCCL's "duplicate GPU" check:
    # ncclInvalidUsage at communicator set-up, before any kernel), both ranks must report that same refusal and
    # the test records it -- a reduce between two physical GPUs needs a larger machine.
    import sys
    helper = os.path.join(ROOT, "tests", "gpu_two_rank_rccl.py")
    env = dict(os.environ, NCCL_DEBUG="WARN")
    procs = [subprocess.Popen([sys.executable, helper, str(k), str(tmp_path)], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True, env=env) for k in range(2)]
    outs = []
    try:
        for p in procs:
            out, err = p.communicate(timeout=150)
            outs.append((p.returncode, out.strip().splitlines()[-1] if out.strip() else "", err[-2000:]))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    print("two-rank RCCL on one device:", [o[:2] for o in outs])
    assert all(rc == 0 for rc, _, _ in outs), outs
    if all(line == "ok" for _, line, _ in outs):
        sc = pa.load_ron_scene(os.path.join(FIXTURES, "scene1.ron"))
        with pa.Renderer(sc, pa.Settings(1, 50), 128, 72) as r:
            r.render(10)
            one = r.read_accum()
        acc = np.load(tmp_path / "acc.npy")
        assert np.all(acc[..., 3] == 1.0)
        rel = np.abs(acc[..., :3] - one[..., :3]) / np.maximum(1.0, np.abs(one[..., :3]))
        assert rel.max() <= 2e-6
    else:
        assert all(line.startswith("init_error 5 ") for _, line, _ in outs), outs     # ncclInvalidUsage on both ranks
