"""mi355pt_render_static (rendering without any device allocation, results in the library's static workspace)
against mi355pt_render, through the C ABI, bit for bit."""
import os

import numpy as np
import pytest

from conftest import FIXTURES

pytestmark = pytest.mark.gpu


def bits_equal(a, b):
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def both_ways(pa, scene, settings, W, H, seeds, **kw):
    with pa.Renderer(scene, settings, W, H, **kw) as r:
        r.render(seeds=seeds)
        want, st_want = r.read_accum(), r.stats()
    with pa.Renderer(scene, settings, W, H, **kw) as r:
        r.render_static(seeds=seeds)
        got, st_got = r.read_accum(), r.stats()
        assert r.samples() == len(seeds)
    assert bits_equal(want, got)
    assert st_got["rays"] == st_want["rays"] and st_got["paths"] == st_want["paths"]
    return want


def test_static_workspace_render_equals_render(pa):
    sc = pa.load_ron_scene(os.path.join(FIXTURES, "scene1.ron"))
    seeds = np.arange(1, 41, dtype=np.uint32) * np.uint32(2654435761)
    # two calls in a row on one device, the second smaller: the workspace keeps the first call's contents
    big = both_ways(pa, sc, pa.Settings(1, 50), 640, 360, seeds)          # 40 frames, several launches' worth at most
    small = both_ways(pa, sc, pa.Settings(1, 50), 100, 57, seeds[:3])     # ragged tiles, 3 frames
    assert np.isfinite(big).all() and np.isfinite(small).all()


def test_static_workspace_same_context_mixed_calls(pa):
    # render, render_static and render again on ONE context continue the same accumulator and seed schedule
    sc = pa.load_ron_scene(os.path.join(FIXTURES, "scene3.ron"))
    W, H = 200, 112
    with pa.Renderer(sc, pa.Settings(2, 50), W, H) as r:
        r.render(9)
        want = r.read_accum()
    with pa.Renderer(sc, pa.Settings(2, 50), W, H) as r:
        r.render(4)
        r.render_static(3)
        r.render(1, wait=False)
        r.render_static(1)          # joins the queued frame first
        assert r.samples() == 9
        assert bits_equal(want, r.read_accum())


def test_static_workspace_scene_in_global_memory(pa):
    # a scene walked from global memory: the stack spill area comes out of the workspace too
    from pathtracer_amd import synth
    sc = synth.plane_scene(30000)
    seeds = [11, 12, 13]
    both_ways(pa, sc, pa.Settings(1, 50), 160, 90, seeds)
    both_ways(pa, sc, pa.Settings(1, 50), 64, 36, seeds[:1])


def test_static_workspace_limits(pa):
    sc = pa.load_ron_scene(os.path.join(FIXTURES, "scene2.ron"))
    with pa.Renderer(sc, pa.Settings(1, 50), 64, 36) as r:
        with pytest.raises(pa.Mi355ptError, match="at most 1024 frames"):
            r.render_static(1025)
        r.render_static(0)
        assert r.samples() == 0
    # 8192 x 8192: one frame of results (805 MB) is beyond the 160 MiB workspace
    with pa.Renderer(sc, pa.Settings(1, 50), 8192, 8192) as r:
        with pytest.raises(pa.Mi355ptError, match="static workspace"):
            r.render_static(1)
        r.resize(3840, 2160)      # fits
        r.render_static(seeds=[5])
        a = r.read_accum()
        r.clear()
        r.render(seeds=[5])
        assert bits_equal(a, r.read_accum())


def test_static_workspace_call_allocates_no_device_memory(pa):
    # free device memory as the HIP runtime reports it is the same before and after a render_static call (after
    # one warm-up call: the first launch of a kernel may load its code object); a plain render of the same frames
    # in a fresh context grows the context's buffers and shows up in the same reading
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")

    def free_bytes():
        free, total = C.c_size_t(0), C.c_size_t(0)
        assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
        return free.value

    sc = pa.load_ron_scene(os.path.join(FIXTURES, "scene1.ron"))
    seeds = np.arange(7, 19, dtype=np.uint32)
    with pa.Renderer(sc, pa.Settings(1, 50), 640, 360) as r:
        r.render_static(seeds=seeds[:2])           # warm-up
        before = free_bytes()
        r.render_static(seeds=seeds)
        r.render_static(seeds=seeds[:5])
        assert free_bytes() == before
        r.render(seeds=seeds)                      # control: this one allocates a frame buffer and a seed ring
        assert free_bytes() < before
