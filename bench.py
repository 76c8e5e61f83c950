#!/usr/bin/env python3
"""Benchmark of the raster/shade hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W [--mode frames|sortfirst] [--config 3]

Workload (config 3 of BASELINE.json): 10,000 instances of a 1,024-triangle mesh = 10.24 M
triangles, 3840x2160, with the per-frame ECS rotation update running on the device.  One step is
one pass of the hot path: rotation system -> camera -> draw of one frame, inputs resident in HBM.

N > 1 (one process per GPU, launched by torch.distributed.run):
  frames     (default) frame-parallel: rank r renders animation frames r, r+N, ...; every rank
             advances the replicated rotation state through all frames; no data-path collective.
             One step = N frames.  Weak scaling.
  sortfirst  each frame is cut into N bands of whole tile rows; geometry is replicated; the
             finished bands are gathered on rank 0 over RCCL.  One step = 1 frame.  Strong scaling.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak
SEED = 3
DELTA_MS = 16.0


def algorithmic_bytes(stats, width, height, covered_px, texture_bytes):
    """Per-kernel algorithmic bytes of one frame.

    setup (k_setup does the vertex stage and the triangle setup in one launch):
        48 B per instance vertex + 60 B per submitted triangle + 32 B per surviving triangle
    raster_shade: 32 B per surviving triangle + 8 B per pixel + texel lower bound
    object: one 64-B model matrix in, one 64-B PVM out per object
    """
    t, tv, vi = stats["prims"], stats["records"], stats["instance_vertices"]
    texel = min(covered_px * 16, texture_bytes)
    return {
        "setup": vi * 48 + t * 60 + tv * 32,
        "raster_shade": tv * 32 + width * height * 8 + texel,
        "object": stats["objects"] * (64 + 64),
    }


def cpu_baseline(scene, width, height, n_prims):
    """The CPU oracle (band-parallel variant) on one full frame of the same workload."""
    from oracle import oracle
    oracle.build()
    cores = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")), 16)
    models = oracle.to_model(scene.transforms)
    view, proj = oracle.camera(scene.camera, width, height)
    t0 = time.perf_counter()
    frames = 0
    while True:
        oracle.render(scene, models, view, proj, width, height, threads=cores)
        frames += 1
        dt = time.perf_counter() - t0
        if dt > 12.0 or frames >= 64:
            break
    return {"value": n_prims * frames / dt, "unit": "triangles/s", "cores": cores, "kind": "port",
            "sample": "%d full frame(s) of the same scene and camera (frame 0 pose), %.1f s of CPU work, "
                      "band-parallel oracle with %d threads" % (frames, dt, cores),
            "ms_per_frame": 1e3 * dt / frames}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mode", choices=["frames", "sortfirst"], default="frames")
    ap.add_argument("--config", type=int, default=3)
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the workload (debug only; marks the line invalid)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--check", action="store_true", help="compare the last frame with the oracle (slow)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (one process per GPU)" % args.gpus)
        args.gpus = world

    import torch
    import torch.distributed as dist
    from renderer_amd import api, scenes, shard

    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: no HIP device is visible and there is no CPU path")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    scene = scenes.config_scene(args.config, scale=args.scale)
    width, height = scene.size
    n_prims = scene.n_triangles
    sortfirst = args.mode == "sortfirst" and world > 1
    scissor = shard.band_rows(height, world, rank) if sortfirst else None

    drawer = api.HipDrawer(width, height, device=local_rank, n_images=2, scissor=scissor, timing=True)
    drawer.update_scene(scene)
    drawer.update_camera(scene.camera)

    frame_no = [0]

    def step(i):
        # ECS rotation update on the device, then the frame
        if sortfirst or world == 1:
            drawer.rotation_system(SEED, frame_no[0], DELTA_MS)
            frame_no[0] += 1
        else:
            # frame-parallel: this rank renders frame (i*world + rank); the updates of the frames
            # the other ranks render are applied in the same launch
            target = i * world + rank
            if frame_no[0] <= target:
                drawer.rotation_system(SEED, frame_no[0], DELTA_MS, n_frames=target - frame_no[0] + 1)
                frame_no[0] = target + 1
        drawer.update_camera(scene.camera)
        drawer.draw(i & 1)
        if sortfirst:
            drawer.sync()
            y0, y1 = scissor
            shard.gather_bands(shard.band_tensor(drawer, i & 1, y0, y1), height, width, world, rank)

    def fence():
        drawer.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for i in range(args.warmup):
        step(i)
    fence()
    drawer.reset_timing_totals()
    # events around every kernel cost about 8 % of a frame: sample every fourth frame of the timed
    # region instead of all of them
    timing_stride = 4 if args.steps >= 16 else 1
    drawer.set_timing_stride(timing_stride)
    t0 = time.perf_counter()
    for i in range(args.steps):
        # frames are enqueued back to back (the library keeps up to three in flight on its
        # stream); nothing waits on the host until the closing fence
        step(args.warmup + i)
    fence()
    elapsed = time.perf_counter() - t0
    # per-kernel device times: HIP events the library records on its own stream around every
    # kernel of the sampled frames of the timed region
    acc, timed_frames = drawer.timing_totals()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    stats = drawer.stats()
    kern = {k: v / max(timed_frames, 1) for k, v in acc.items()}
    frames_per_step = 1 if (sortfirst or world == 1) else world
    total_frames = frames_per_step * args.steps
    ms_per_step = 1e3 * elapsed / max(args.steps, 1)
    tri_per_s = n_prims * total_frames / elapsed
    fps = total_frames / elapsed

    result = None
    if rank == 0:
        cov = drawer.read_depth((args.warmup + args.steps - 1) & 1)
        covered_px = int((cov < 1.0).sum())
        tex_bytes = sum(int(t.nbytes) for t in scene.textures)
        ab = algorithmic_bytes(stats, width, height if not sortfirst else (scissor[1] - scissor[0]), covered_px, tex_bytes)
        stage_ms = {"setup": kern["setup_ms"], "raster_shade": kern["raster_shade_ms"]}
        dom = max(stage_ms, key=stage_ms.get)
        achieved = ab[dom] / (stage_ms[dom] * 1e-3) / 1e9 if stage_ms[dom] > 0 else 0.0
        traffic = None
        pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc_path):
            try:
                traffic = json.load(open(pmc_path)).get(dom)
            except Exception:
                traffic = None
        result = {
            "metric": "triangles_per_sec", "value": tri_per_s, "unit": "triangles/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong" if sortfirst else "weak", "vs_baseline": None, "dtype": "f32+i64",
            "data": "synthetic",
            "frames_per_sec": fps,
            "config": {"workload": "config 3: %s, %d objects x 1024 triangles = %d triangles, %dx%d, "
                                   "per-frame device-side rotation update" % (scene.name, len(scene.objects), n_prims, width, height),
                       "parallelism": ("sortfirst-bands%d+rccl-gather" % world) if sortfirst else
                                      ("frame-parallel%d" % world if world > 1 else "single"),
                       "frames_per_step": frames_per_step, "tile": stats["tile_size"]},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": ab[dom], "avg_ms_per_launch": stage_ms[dom]},
            # every stage against the same HBM peak, and the whole frame's algorithmic bytes over
            # the sum of the kernel times (the dominant kernel is the one reported in "roofline")
            "stage_rooflines": {k: {"algorithmic_bytes": ab[k], "avg_ms": stage_ms[k],
                                    "achieved_GBps": ab[k] / (stage_ms[k] * 1e-3) / 1e9 if stage_ms[k] > 0 else 0.0}
                                for k in stage_ms},
            "frame_roofline": {"algorithmic_bytes": sum(ab.values()), "frame_ms": kern.get("frame_ms", 0.0),
                               "achieved_GBps": sum(ab.values()) / (kern["frame_ms"] * 1e-3) / 1e9 if kern.get("frame_ms") else 0.0},
            "kernels_ms": kern,
            "kernel_timing": {"frames_sampled": int(timed_frames), "stride": timing_stride},
            "frame_stats": stats,
        }
        if args.scale != 1.0:
            result["invalid"] = "scaled-down debug run"
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(scene, width, height, n_prims)
        if args.check:
            from oracle import oracle
            t = drawer.read_transforms()
            ref = oracle.render(scene, oracle.to_model(t), *oracle.camera(scene.camera, width, height), width, height,
                                threads=16)
            img = (args.warmup + args.steps - 1) & 1
            result["check"] = {"color_equal": bool(np.array_equal(ref["color"], drawer.read_color(img))),
                               "depth_equal": bool(np.array_equal(ref["depth"].view(np.uint32),
                                                                  drawer.read_depth(img).view(np.uint32)))}
    drawer.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
