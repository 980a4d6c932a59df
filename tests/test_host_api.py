"""Host side of the library (no GPU needed): symbol exports, RON loader, camera, reference BVH,
seed schedule, image post-process and PNG -- each against the oracle or an independent reader."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import FIXTURES, ROOT, bits_equal, oracle_scene


def test_library_exports_every_declared_symbol(pa):
    hdr = open(os.path.join(ROOT, "include", "mi355pt.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(mi355pt_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 30
    L = C.CDLL(pa.library_path())
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in mi355pt.h but not exported"
    from pathtracer_amd import binding
    assert declared == set(binding.SIGNATURES), "binding.SIGNATURES out of sync with the header"


def test_rust_declarations_cover_the_header(pa):
    # the -sys crate cannot be compiled here (no rustc); at least keep its names and struct fields in step
    hdr = open(os.path.join(ROOT, "include", "mi355pt.h")).read()
    hdr_nc = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(mi355pt_[a-z0-9_]+)\s*\(", hdr_nc))
    rs = open(os.path.join(ROOT, "bindings", "rust", "mi355pt-sys", "src", "lib.rs")).read()
    rust_fns = set(re.findall(r"pub fn (mi355pt_[a-z0-9_]+)\s*\(", rs))
    assert rust_fns == declared
    stats_c = re.search(r"typedef struct mi355pt_stats \{(.*?)\} mi355pt_stats;", hdr_nc, re.S).group(1)
    fields_c = re.findall(r"\b([a-z_]+)(?:\[\d+\])?;", stats_c)
    stats_rs = re.search(r"pub struct mi355pt_stats \{(.*?)\n\}", rs, re.S).group(1)
    fields_rs = re.findall(r"pub ([a-z_]+):", stats_rs)
    assert fields_rs == fields_c
    from pathtracer_amd import binding
    assert [f[0] for f in binding.Stats._fields_] == fields_c


def test_library_is_hip_only(pa):
    # the product links the HIP runtime and exposes no oracle entry points
    out = os.popen(f"readelf -d {pa.library_path()}").read()
    assert "libamdhip64" in out
    syms = os.popen(f"nm -D --defined-only {pa.library_path()}").read()
    assert "orc_" not in syms


@pytest.mark.parametrize("name", ["scene1", "scene2", "scene3"])
def test_ron_loader_matches_independent_reader(pa, orc, name):
    path = os.path.join(FIXTURES, name + ".ron")
    sc = pa.load_ron_scene(path)
    ref = orc.ronlite.scene_blobs(orc.ronlite.load(path))
    assert sc is not None
    assert sc.spheres.tobytes() == ref["spheres"].tobytes()
    assert sc.lights.tobytes() == ref["lights"].tobytes()
    assert sc.lambertians.tobytes() == ref["lambertians"].tobytes()
    assert sc.metals.tobytes() == ref["metals"].tobytes()
    assert sc.glass.tobytes() == ref["glass"].tobytes()
    pos, yaw, pitch, vfov = ref["camera"]
    assert tuple(sc.camera.pos) == tuple(pos) and sc.camera.yaw == yaw and sc.camera.pitch == pitch
    assert sc.camera.vfov == vfov


def test_scene1_material_mix(pa):
    sc = pa.load_ron_scene(os.path.join(FIXTURES, "scene1.ron"))
    counts = np.bincount(sc.spheres["mat_type"], minlength=4)
    assert list(counts) == [89, 80, 63, 88] and len(sc.spheres) == 320


def test_ron_errors_return_none_like_load_ron(pa, tmp_path):
    assert pa.load_ron_scene(tmp_path / "missing.ron") is None
    bad = tmp_path / "bad.ron"
    bad.write_text("(camera: (pos: (x: 0, y: 0, z: 0), yaw: 0, pitch: 0, vfov: 45), spheres: [")
    assert pa.load_ron_scene(bad) is None
    nofield = tmp_path / "nofield.ron"
    nofield.write_text("(camera: (pos: (x: 0.0, y: 0.0, z: 0.0), yaw: 0.0, pitch: 0.0, vfov: 45.0), spheres: [], "
                       "lights: [], lambertians: [], metals: [])")  # glass missing
    assert pa.load_ron_scene(nofield) is None
    assert pa.load_ron_settings(tmp_path / "missing.ron") is None
    s = pa.load_ron_settings(os.path.join(FIXTURES, "settings.ron"))
    assert (s.samples, s.depth) == (1, 50)
    floaty = tmp_path / "floaty.ron"
    floaty.write_text("(samples: 1.5, depth: 50)")
    assert pa.load_ron_settings(floaty) is None


def test_ron_syntax_variants(pa):
    text = """Scene( // struct name and comments are allowed
        camera: (pos: (x: 1, y: -0.0, z: 2.5e0), yaw: 10.0, pitch: -5.0, vfov: 30.0,),
        spheres: [ (pos: (x: 0.0, y: 0.0, z: 0.0), rad: 1.0, mat_type: 1, mat_index: 0),
                   (pos: (x: 3.0, y: 0.0, z: 0.0), rad: 0.5, mat_type: 0, mat_index: 0), ],
        lights: [(colour: (x: 1.0, y: 0.5, z: 0.25))], /* block comment */
        lambertians: [(albedo: (x: 0.1, y: 0.2, z: 0.3),),],
        metals: [], glass: [],
        unknown_field: 3,
    )"""
    sc = pa.Scene.from_ron_text(text)
    assert len(sc.spheres) == 2 and len(sc.metals) == 0 and len(sc.glass) == 0
    assert tuple(sc.camera.pos) == (1.0, -0.0, 2.5) and np.signbit(sc.camera.pos[1])
    assert list(sc.lights["v"][0]) == [1.0, 0.5, 0.25]


def test_default_scene(pa, orc):
    # scene.rs:15-31 / camera.rs:13-22
    sc = pa.Scene.default()
    ref = orc.ronlite.scene_blobs(orc.ronlite.default_scene())
    assert sc.spheres.tobytes() == ref["spheres"].tobytes()
    assert sc.metals.tobytes() == ref["metals"].tobytes() and sc.glass.tobytes() == ref["glass"].tobytes()
    assert tuple(sc.camera.pos) == (0.0, 0.0, -3.0) and sc.camera.vfov == 60.0


@pytest.mark.parametrize("name,w,h", [("scene1", 1280, 720), ("scene1", 1920, 1080), ("scene2", 400, 225),
                                      ("scene3", 333, 77)])
def test_camera_uniform_bytes(pa, orc, name, w, h):
    sc = pa.load_ron_scene(os.path.join(FIXTURES, name + ".ron"))
    osc = oracle_scene(orc, sc)
    aspect = np.float32(w) / np.float32(h)
    got = pa.camera_build(sc.camera.pos, sc.camera.yaw, sc.camera.pitch, sc.camera.vfov, aspect)
    assert bits_equal(got, osc.camera(w, h))
    assert got[3] == got[7] == got[11] == got[15] == 0.0  # the 4-byte pads


@pytest.mark.parametrize("swap_mode", [0, 1, 2])
def test_reference_bvh_bytes(pa, orc, swap_mode):
    for name in ("scene1", "scene2", "scene3"):
        sc = pa.load_ron_scene(os.path.join(FIXTURES, name + ".ron"))
        a = pa.bvh_build_reference(sc.spheres, swap_mode, 0xBEEF)
        b = orc.bvh_build(sc.spheres, swap_mode, 0xBEEF)
        assert a.tobytes() == b.tobytes()
    rng = np.random.default_rng(3)
    for n in (2, 3, 5, 6, 7, 9, 31, 257):
        sp = np.zeros(n, pa.renderer.SPHERE_DT)
        sp["pos"] = rng.normal(0, 4, (n, 3)).astype(np.float32)
        sp["pos"][::3, 2] = 1.0  # equal sort keys exercise the stable sort
        sp["rad"] = rng.uniform(0.1, 1, n).astype(np.float32)
        assert pa.bvh_build_reference(sp, swap_mode, 5).tobytes() == orc.bvh_build(sp, swap_mode, 5).tobytes()


def test_reference_bvh_rejects_what_the_reference_panics_on(pa):
    with pytest.raises(pa.Mi355ptError):
        pa.bvh_build_reference(np.zeros(1, pa.renderer.SPHERE_DT))
    sp = np.zeros(4, pa.renderer.SPHERE_DT)
    sp["pos"][2, 2] = np.nan
    with pytest.raises(pa.Mi355ptError):
        pa.bvh_build_reference(sp)


def test_frame_seed_schedule(pa):
    seeds = [pa.frame_seed(0x00C0FFEE, f) for f in range(4096)]
    assert len(set(seeds)) == 4096
    assert seeds[:4] == [pa.frame_seed(0x00C0FFEE, f) for f in range(4)]  # pure function
    assert pa.frame_seed(0x00C0FFEE, 0) != pa.frame_seed(0x00C0FFEF, 0)

    def ref(base, f):  # documented mixer: murmur3 finaliser of base ^ (f * golden + c)
        x = (base ^ ((f * 0x9E3779B9 + 0x7F4A7C15) & 0xFFFFFFFF)) & 0xFFFFFFFF
        x ^= x >> 16
        x = (x * 0x85EBCA6B) & 0xFFFFFFFF
        x ^= x >> 13
        x = (x * 0xC2B2AE35) & 0xFFFFFFFF
        x ^= x >> 16
        return x
    for base, f in ((0, 0), (0x00C0FFEE, 7), (0xFFFFFFFF, 0xFFFFFFFF), (123, 456789)):
        assert pa.frame_seed(base, f) == ref(base, f)


def test_postprocess_matches_oracle_and_formula(pa, orc):
    rng = np.random.default_rng(0)
    acc = rng.uniform(0, 40, (20, 30, 4)).astype(np.float32)
    acc[0, 0, :3] = (0.0, 1e9, -1.0)   # clamp both ways (powf of a negative is NaN -> 0)
    frames = 16
    got = pa.postprocess_rgb8(acc, frames)
    assert np.array_equal(got, orc.postprocess(acc, frames))
    v = np.power((acc[..., :3] * np.float32(1.0 / frames)).astype(np.float32), np.float32(1 / 2.2))
    want = np.round(np.clip(np.nan_to_num(v, nan=0.0), 0, 1) * 255).astype(np.int32)
    assert np.abs(got.astype(np.int32) - want).max() <= 1


def test_png_roundtrip(pa, tmp_path):
    from PIL import Image
    rng = np.random.default_rng(1)
    rgb = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    p = tmp_path / "img.png"
    pa.write_png_rgb8(p, rgb)
    im = Image.open(p)
    assert im.mode == "RGB" and im.size == (53, 37)
    assert np.array_equal(np.asarray(im), rgb)


def test_create_without_gpu_fails_loudly(pa):
    if os.path.exists("/dev/kfd"):
        pytest.skip("a GPU is present")
    sc = pa.Scene.default()
    with pytest.raises(pa.Mi355ptError) as e:
        pa.Renderer(sc, pa.Settings(), 16, 9)
    assert "no CPU path" in str(e.value)


def test_synthetic_plane_scene_is_deterministic_and_valid(pa):
    from pathtracer_amd import synth
    a = synth.plane_scene(2000)
    b = synth.plane_scene(2000)
    assert a.spheres.tobytes() == b.spheres.tobytes() and a.glass.tobytes() == b.glass.tobytes()
    n = len(a.spheres)
    assert 0.9 * 2000 < n < 1.3 * 2000
    t = a.spheres["mat_type"]
    frac = np.bincount(t[4:], minlength=4) / (n - 4)
    assert np.all(np.abs(frac - 0.25) < 0.05)  # quartiles, gen_rand_scene.rs:79-106
    for arr, k in ((a.lights, 0), (a.lambertians, 1), (a.metals, 2), (a.glass, 3)):
        assert a.spheres["mat_index"][t == k].max() < len(arr)
    assert np.all(a.glass >= 1.0) and np.all(a.glass < 4.0)
    assert np.all(a.metals["w"] <= 0.5)
    assert abs(float(a.camera.yaw) - (-102.99461)) < 1e-3 and abs(float(a.camera.pitch) - (-8.525517)) < 1e-3


def test_frame_partition_is_exact(pa):
    # mi355pt_partition_frames: contiguous blocks that tile [0, total), sizes differ by at most one; the Python
    # helper the multi-process path uses is the same function
    from pathtracer_amd import parallel
    for total in (0, 1, 7, 256, 1000, 16384):
        for parts in (1, 2, 3, 8, 13):
            at, sizes = 0, []
            for k in range(parts):
                first, count = pa.partition_frames(total, parts, k)
                assert first == at
                assert list(parallel.rank_frames(total, k, parts, first=100)) == list(range(100 + first, 100 + first + count))
                at += count
                sizes.append(count)
            assert at == total and max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
    assert pa.partition_frames(10, 0, 0) == (0, 0) and pa.partition_frames(10, 2, 2) == (0, 0)
