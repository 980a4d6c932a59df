#!/usr/bin/env python3
"""Benchmark of the hot path on MI355X: scene1.ron at 1920x1080, 1 sample per frame, depth 50.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step is one pass of the path kernel over one batch of frames (default 256 frames = 530.8 M paths per
GPU, one launch); 4 steps are the 1024 frames of the headline configuration, the default 16 steps render
4096 frames.  With N GPUs every rank
renders its own disjoint frames (weak scaling), and the accumulators are sum-reduced to rank 0 over
RCCL inside the timed region.  Rank 0 prints one JSON line.
"""
import argparse
import json
import re
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_VALU_TFLOPS = 157.3   # MI355X vector FP32 peak (= the FP32 MFMA peak): 256 CUs x 128 FMA lanes x 2.4 GHz
PEAK_HBM_GBS = 8000.0
# rocprofv3 summary of this command (tools/profile.sh): source of roofline.traffic and of the issue-slot figure
PROFILE = os.path.join("profiles", "r03_final", "pmc_summary.json")
# highest sustained vector issue rate measured on MI355X (tools/micro/valu_peak, profiles/r01_micro/valu_peak.txt)
MEASURED_VALU_ISSUE_PER_SIMD = 9.07e8   # wave-instructions / s / SIMD


def flops_per_launch(st):
    """Algorithmic FLOPs of the traversal actually run, from the counting kernels: 24 per box test
    (two per fetched node), 17 per sphere test, 18 per accepted hit, and per bounce 60 (lambertian /
    metal incl. the mean 1.91 rejection tries), 45 (glass), 8 (sky), 0 (light)."""
    b = st["bounces"]
    return (st["nodes"] * 2 * 24 + st["tests"] * 17 + st["accepted"] * 18 +
            (b["lambertian"] + b["metal"]) * 60 + b["glass"] * 45 + b["sky"] * 8)


def bytes_per_launch(st):
    """Algorithmic scene bytes fetched: 64 per interior node record, 16 per sphere record, 24 per
    bounce (sphere + material info + material record)."""
    hits = sum(st["bounces"][k] for k in ("light", "lambertian", "metal", "glass"))
    return st["nodes"] * 64 + st["tests"] * 16 + hits * 40


# The other configurations of the benchmark set, measured after the timed region on one GPU (reported under
# "configs"): name -> (scene file or sphere count of the synthetic plane, width, height, frames).  The
# reference's harness for all of them is the one loop of src/bin/offline.rs:44-52.
CONFIGS = {
    "c1_scene1_400x225x16": ("scene1.ron", 400, 225, 16),
    "c3_scene2_1080p_x256": ("scene2.ron", 1920, 1080, 256),
    "c3_scene3_1080p_x256": ("scene3.ron", 1920, 1080, 256),
    "c4_1M_spheres_4k_x256": (1_000_000, 3840, 2160, 256),
    "c5_10M_spheres_4k_x64_one_gpu": (10_000_000, 3840, 2160, 64),
}


def measure_config(pa, src, W, H, F, device=0, bvh_mode=None, count_frames=4, repeats=3):
    """One configuration on one GPU: F frames in one call, `repeats` times after a warm-up that also measures the
    tile order, the median call reported (the first call after a pause runs at lower clocks); kernel time from the
    library's HIP events; visit counts from the instrumented kernels on a few frames."""
    from pathtracer_amd import synth
    t0 = time.perf_counter()
    sc = pa.load_ron_scene(os.path.join(ROOT, "fixtures", src)) if isinstance(src, str) else synth.plane_scene(src)
    t_scene = time.perf_counter() - t0
    kw = {} if bvh_mode is None else {"bvh_mode": bvh_mode}
    t0 = time.perf_counter()
    with pa.Renderer(sc, pa.Settings(1, 50), W, H, device=device, **kw) as r:
        t_create = time.perf_counter() - t0
        r.render(min(F, 16))
        runs = []
        for _ in range(max(1, repeats)):
            r.clear()
            t0 = time.perf_counter()
            r.render(F)
            wall = time.perf_counter() - t0
            s = r.stats()
            runs.append((s["render_ms"], wall, s))
        runs.sort(key=lambda x: x[0])
        _, wall, s = runs[len(runs) // 2]
    with pa.Renderer(sc, pa.Settings(1, 50), W, H, device=device, flags=pa.FLAG_COUNT, **kw) as rc:
        rc.render(min(F, count_frames))
        c = rc.stats()
    k_ms = s["render_ms"]
    fl, by = flops_per_launch(c) / c["rays"], bytes_per_launch(c) / c["rays"]
    return {"spheres": int(len(sc.spheres)), "size": [W, H], "frames": F,
            "Mrays_s": s["rays"] / k_ms / 1e3, "Mpaths_s": s["paths"] / k_ms / 1e3,
            "Mrays_s_with_accumulation": s["rays"] / (k_ms + s["resolve_ms"]) / 1e3,
            "kernel_ms": k_ms, "resolve_ms": s["resolve_ms"], "launches": s["launches"], "wall_ms": wall * 1e3,
            "rays_per_path": s["rays"] / s["paths"],
            "nodes_per_ray": c["nodes"] / c["rays"], "tests_per_ray": c["tests"] / c["rays"],
            "flops_per_ray": fl, "bytes_per_ray": by,
            "TFLOPs": fl * s["rays"] / k_ms / 1e9, "frac_valu": fl * s["rays"] / k_ms / 1e9 / PEAK_FP32_VALU_TFLOPS,
            "scene_fetch_GBps": by * s["rays"] / k_ms / 1e6,
            "scene_in_lds": bool(s["scene_in_lds"]), "scene_MB": s["scene_bytes"] / 1e6,
            "bvh_nodes": s["bvh_nodes"], "bvh_depth": s["bvh_depth"],
            "camera_ray_list_tiles": s["list_tiles"], "sky_tiles": s["sky_tiles"], "tiles": ((W + 7) // 8) * ((H + 7) // 8),
            "repeats": len(runs), "kernel_ms_all": [x[0] for x in runs],
            "scene_s": t_scene, "create_s": t_create}


def profile_counters(kernel_prefix, instance=r"<\d+, false,"):
    """Per-launch counter averages of the timed (non-counting) instance of the dominant kernel from the
    committed rocprofv3 passes of this same command, or None."""
    try:
        d = json.load(open(os.path.join(ROOT, PROFILE)))
    except Exception:
        return None
    best, most = None, 0
    for name, v in d.items():
        if kernel_prefix in name and (instance is None or re.search(instance, name)):   # <SCENE, COUNT = false, lanes, ...>
            c = v.get("counters_avg_per_dispatch", {})
            n = max((x["dispatches"] for x in c.values()), default=0)
            if n > most:   # the instance most launches run (the cost-measuring launches run another)
                best, most = {k: x["avg"] for k, x in c.items()}, n
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames-per-step", type=int, default=256)
    ap.add_argument("--total-frames", type=int, default=0,
                    help="strong scaling: every step renders this many frames in total, split over the ranks "
                         "(default 0: weak scaling, --frames-per-step frames per rank and step)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--scene", default=os.path.join(ROOT, "fixtures", "scene1.ron"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-rmse", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the other configurations (reported under 'configs')")
    args = ap.parse_args()

    # stdout carries the one JSON line and nothing else: what libraries print there while they start up (RCCL's
    # version banner, gloo's connection lines) goes to stderr; the descriptor comes back for the final print
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch  # first, so the process uses one HIP runtime for torch and the library
    import torch.distributed as dist
    import numpy as np

    import pathtracer_amd as pa
    from pathtracer_amd import parallel

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the library has no CPU path)")
    # MI355PT_BENCH_BACKEND=gloo rehearses the multi-rank flow on a box with fewer GPUs than ranks
    # (ranks then share devices and the reduce is staged through the host); the measured path is nccl.
    backend = os.environ.get("MI355PT_BENCH_BACKEND", "nccl")
    if backend == "gloo":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    # MI355PT_BENCH_DIST=1 under a one-rank torchrun takes the multi-rank code path (process group, reduce,
    # all-reduces) with a world of one: the only RCCL rehearsal a one-GPU box allows
    use_dist = world > 1 or os.environ.get("MI355PT_BENCH_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    W, H, F = args.width, args.height, args.frames_per_step
    scene = pa.load_ron_scene(args.scene)
    settings = pa.Settings(1, 50)
    base_seed = 0x00C0FFEE
    r = pa.Renderer(scene, settings, W, H, device=local_rank, base_seed=base_seed)
    acc_t = parallel.accum_tensor(r, local_rank) if use_dist else None

    strong = args.total_frames > 0
    step_frames = args.total_frames if strong else F * world   # frames all ranks render per step

    def step_seeds(step, rank=rank, world=world):
        # frames [step * step_frames, (step + 1) * step_frames) of one seed schedule, split over the ranks by
        # mi355pt_partition_frames (contiguous blocks): weak scaling gives every rank F of them
        return parallel.seeds_for(base_seed, parallel.step_frames(step, F, rank, world, args.total_frames), pa.frame_seed)

    def sync():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    for s in range(args.warmup):
        r.render(seeds=step_seeds(args.steps + s))
    if use_dist and args.warmup:
        parallel.reduce_accum(torch.zeros_like(acc_t), world, dst=0)  # communicator set-up outside the timed region
    r.clear()
    sync()
    t0 = time.perf_counter()
    for s in range(args.steps):
        r.render(seeds=step_seeds(s), wait=False)   # a step's accumulation overlaps the next step's path kernel
    r.wait()
    if use_dist:
        parallel.reduce_accum(acc_t, world, dst=0)
    sync()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    st = r.stats()
    counts = torch.tensor([st["paths"], st["rays"]], dtype=torch.float64, device="cuda")
    if use_dist:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    paths_all, rays_all = float(counts[0].item()), float(counts[1].item())

    if rank == 0:
        kernel_ms = st["render_ms"] / max(1, st["launches"])   # HIP events on the render stream
        out = {
            "metric": "Mrays/s", "value": rays_all / dt / 1e6, "unit": "Mrays/s",
            "mpaths_per_s": paths_all / dt / 1e6,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt * 1e3 / max(1, args.steps), "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "f32",
            "data": f"the reference's own scene file {os.path.basename(args.scene)}; frame seeds from a fixed schedule (base 0x00C0FFEE)",
            "config": {"workload": (f"{os.path.basename(args.scene)} {W}x{H}, {step_frames} frames/step over all GPUs x "
                                    f"{args.steps} steps, 1 spp/frame, depth 50" if strong else
                                    f"{os.path.basename(args.scene)} {W}x{H}, {F} frames/step x {args.steps} steps "
                                    f"per GPU, 1 spp/frame, depth 50"),
                       "frames_per_step_all_gpus": step_frames,
                       "frames_per_gpu": (step_frames * args.steps) // world if strong else F * args.steps,
                       "paths_per_step_per_gpu": W * H * (step_frames // world),
                       "parallelism": f"frame-partitioned x{world}, one {'RCCL' if backend == 'nccl' else backend} sum-reduce"
                                      if world > 1 else "1 GPU",
                       "bvh": "binned SAH, 64-B two-child nodes", "scene_in_lds": bool(st["scene_in_lds"])},
            "rays_per_path": rays_all / paths_all,
            "kernel_ms_per_launch": kernel_ms,
            "resolve_ms_per_launch": st["resolve_ms"] / max(1, st["launches"]),
        }
        # ---- roofline of the dominant kernel (pt_render_kernel), per launch
        with pa.Renderer(scene, settings, W, H, device=local_rank, base_seed=base_seed, flags=pa.FLAG_COUNT) as rc:
            rc.render(seeds=step_seeds(0))
            cs = rc.stats()
        fl = flops_per_launch(cs)
        by = bytes_per_launch(cs)
        ach = fl / (kernel_ms * 1e-3) / 1e12
        out["roofline"] = {
            "bound": "valu", "achieved": ach, "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s",
            "frac": ach / PEAK_FP32_VALU_TFLOPS, "traffic": None, "traffic_unit": "bytes/launch",
            "kernel": "pt_render_sched_kernel", "flops_per_ray": fl / cs["rays"], "rays_per_launch": cs["rays"],
            "nodes_per_ray": cs["nodes"] / cs["rays"], "sphere_tests_per_ray": cs["tests"] / cs["rays"],
            "note": f"scene ({cs['scene_bytes'] / 1024:.0f} KB) is LDS-resident: bounded by vector + scalar instruction issue (see valu_issue), not HBM; peak is the "
                    "vector FP32 figure (equal to the FP32 MFMA peak)",
            "scene_fetch": {"bytes_per_ray": by / cs["rays"], "achieved_GBps": by / (kernel_ms * 1e-3) / 1e9,
                            "served_from": "LDS" if cs["scene_in_lds"] else "L2/HBM", "hbm_peak_GBps": PEAK_HBM_GBS},
        }
        pc = profile_counters("pt_render_sched_kernel")
        if pc and os.path.basename(args.scene) == "scene1.ron" and (W, H, F) == (1920, 1080, 256):
            # HBM bytes per launch: 2 x FETCH_SIZE + WRITE_SIZE, KiB -> bytes (FETCH_SIZE under-reports wide
            # reads by 2x on gfx950)
            tr = (2.0 * pc["FETCH_SIZE"] + pc["WRITE_SIZE"]) * 1024.0
            out["roofline"]["traffic"] = tr
            out["roofline"]["traffic_source"] = PROFILE
            out["roofline"]["hbm_GBps_from_traffic"] = tr / (kernel_ms * 1e-3) / 1e9
            rc_ = profile_counters("pt_accumulate_kernel", None) or profile_counters("pt_resolve_kernel", None)
            if rc_:   # the per-path results are read back once by the accumulation kernel
                out["roofline"]["traffic_with_accumulation_kernel"] = tr + (2.0 * rc_["FETCH_SIZE"] + rc_["WRITE_SIZE"]) * 1024.0
            out["roofline"]["algorithmic_hbm_bytes"] = 2.0 * 16.0 * W * H   # one read-modify-write of the accumulator
            simds = 4 * torch.cuda.get_device_properties(local_rank).multi_processor_count
            rate = pc["SQ_INSTS_VALU"] / simds / (kernel_ms * 1e-3)
            out["roofline"]["valu_issue"] = {
                "wave_instr_per_launch": pc["SQ_INSTS_VALU"], "per_simd_per_s": rate,
                "measured_peak_per_simd_per_s": MEASURED_VALU_ISSUE_PER_SIMD,
                "frac": rate / MEASURED_VALU_ISSUE_PER_SIMD,
                "active_lanes_per_instr": pc["SQ_THREAD_CYCLES_VALU"] / pc["SQ_INSTS_VALU"],
                # the scalar unit is shared by the four SIMDs of a CU: with two vector instructions per scalar
                # one, six waves per SIMD sustain 0.327 vector instructions per cycle instead of 0.38
                # (tools/micro/valu_peak, profiles/r01_micro/valu_peak.txt)
                "scalar_instr_per_launch": pc.get("SQ_INSTS_SALU"),
                "vector_per_scalar": pc["SQ_INSTS_VALU"] / pc["SQ_INSTS_SALU"] if pc.get("SQ_INSTS_SALU") else None,
                "measured_peak_at_2_to_1_mix_per_simd_per_s": 0.327 / 0.38 * MEASURED_VALU_ISSUE_PER_SIMD,
            }
        # ---- converged-image parity against the reference's own render (outside the timed region)
        if not args.no_rmse and os.path.basename(args.scene) == "scene1.ron":
            try:
                from PIL import Image
                ref = np.asarray(Image.open(os.path.join(ROOT, "fixtures", "mdimg.png")).convert("RGB")).astype(np.float64)
                with pa.Renderer(scene, settings, 1280, 720, device=local_rank, base_seed=base_seed) as ri:
                    ri.render(1000)
                    img = ri.to_rgb8().astype(np.float64)
                rmse = float(np.sqrt(np.mean((img - ref) ** 2)))
                out["rmse_vs_reference_png_8bit"] = rmse
                out["psnr_vs_reference_png_db"] = float(20 * np.log10(255.0 / rmse))
            except Exception as e:  # the image check must not hide the measurement
                out["rmse_vs_reference_png_8bit"] = f"unavailable: {e}"
        # ---- CPU baseline: the oracle (restatement of the reference algorithm) on a bounded sample
        if not args.no_cpu_baseline and world == 1:
            from oracle import orc
            osc = orc.Scene.from_ron(args.scene)
            threads = min(16, os.cpu_count() or 1)
            n_cpu_frames = 64   # a quarter of step 0 (10-20 s of CPU work)
            seeds = step_seeds(0)[:n_cpu_frames]
            t1 = time.perf_counter()
            cpu_acc, ost = orc.render(osc, osc.camera(W, H), W, H, seeds, threads=threads)
            cdt = time.perf_counter() - t1
            # the same frames on the GPU, outside the timed region: full-size parity against the CPU result, with
            # the default tree and with the reference's tree walked the reference's way
            parity = {"frames": n_cpu_frames, "size": [W, H]}
            for name, mode in (("default_tree", pa.BVH_SAH), ("reference_walk", pa.BVH_REFERENCE)):
                with pa.Renderer(scene, settings, W, H, device=local_rank, bvh_mode=mode) as rp:
                    rp.render(seeds=seeds)
                    differ = (rp.read_accum().view(np.uint32) != cpu_acc.view(np.uint32)).any(axis=2)
                    pst = rp.stats()
                    parity[name] = {"pixels_differing": int(differ.sum()), "rays_equal": bool(pst["rays"] == ost["rays"]),
                                    "mrays_per_s": pst["rays"] / pst["render_ms"] / 1e3}
            out["parity_vs_cpu_baseline"] = parity
            out["cpu_baseline"] = {
                "value": ost["rays"] / cdt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
                "mpaths_per_s": ost["paths"] / cdt / 1e6,
                "sample": f"{n_cpu_frames} of the step-0 frames of the same workload ({W}x{H}, depth 50), CPU "
                          f"restatement of the reference algorithm (reference BVH topology, 48-B nodes), "
                          f"{cdt:.1f} s wall",
            }
            # The work the reference's own algorithm does for these same rays (its z-split tree walked without
            # ordering or pruning, counted by the CPU restatement): 24 flop per box test, 17 per sphere test, 18
            # per accepted candidate, scatter as in flops_per_launch.  At the GPU's ray rate this is the arithmetic
            # rate a kernel running the reference's walk would need for the same throughput.
            b_om = ost["bounce_lambertian"] + ost["bounce_metal"]
            ref_fl = (ost["interior_visits"] * 24 + ost["leaf_visits"] * 17 + ost["accepted"] * 18 + b_om * 60 +
                      ost["bounce_glass"] * 45 + ost["misses"] * 8) / ost["rays"]
            out["roofline"]["reference_algorithm"] = {
                "flops_per_ray": ref_fl, "interior_visits_per_ray": ost["interior_visits"] / ost["rays"],
                "leaf_visits_per_ray": ost["leaf_visits"] / ost["rays"],
                "equivalent_TFLOPs_at_measured_ray_rate": ref_fl * rays_all / dt / 1e12,
                "equivalent_frac_of_peak": ref_fl * rays_all / dt / 1e12 / PEAK_FP32_VALU_TFLOPS,
            }
        # ---- the other configurations, one GPU, outside the timed region
        if not args.no_configs and world == 1:
            cfgs = {}
            for name, (src, cw, ch, cf) in CONFIGS.items():
                try:
                    cfgs[name] = measure_config(pa, src, cw, ch, cf, device=local_rank)
                except Exception as e:  # a configuration that fails must not hide the measurement
                    cfgs[name] = {"error": str(e)}
            cfgs["note"] = ("kernel time from HIP events, default tree (binned SAH); for the 1 M / 10 M-sphere planes its "
                            "images are within the stated bound of the oracle (<= 0.1 % / 0.3 % of pixels, horizon "
                            "rows), not bit-equal; counters of these kernels: profiles/r03_c4_1M_4k, profiles/r03_c5_10M_4k")
            out["configs"] = cfgs
        sys.stdout.flush()
        os.dup2(json_fd, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    r.close()
    if use_dist:
        dist.barrier()   # rank 0's reporting legs are over: all ranks leave the group together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
