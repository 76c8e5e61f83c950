#!/usr/bin/env python3
"""Benchmark of the raster/shade hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W [--mode auto|sortfirst|frames] [--config 3|4]

Workload (config 3 of BASELINE.json): 10,000 instances of a 1,024-triangle mesh = 10.24 M
triangles, 3840x2160, with the per-frame ECS rotation update running on the device.  One step is
one pass of the hot path: rotation system -> camera -> draw of one frame, inputs resident in HBM.
--config 4 is the same scene at 7680x4320.

N > 1 (one process per GPU, launched by torch.distributed.run):
  frames     frame-parallel: rank r renders animation frames r, r+N, ...; every rank advances the
             replicated rotation state through all frames; no data-path collective.  One step = N
             frames.  Weak scaling: what a frame of 0.2 ms asks for when the goal is frames per second.
  sortfirst  each frame is cut into N bands of whole tile rows; geometry is replicated; the finished
             bands are gathered on rank 0 over RCCL (grouped send/receive, every band at its own
             size) while the next frame is being drawn.  One step = 1 frame.  Strong scaling: what a
             frame asks for when the goal is its latency (the 8K frame of config 4).  Before the
             warm-up the ranks time the drawing of equal bands and their gather on their own; when
             the gather is the slower of the two (few ranks: each band is a large share of the frame
             over one link), rank 0, which sends nothing, takes a taller band (--bands even: equal
             bands).
  auto       (default) frames for config 3, sortfirst for config 4.
  Either way the other split is measured too and reported under a key of its own ("sort_first" or
  "frame_parallel").  The band exchange is the one part of an N-rank run that no single-GPU run
  exercises with its real transport: whatever it raises, and a hang beyond --watchdog-seconds, ends
  up under that key (or, when sort-first is the headline, makes the frame-parallel measurement taken
  first the line of the run) instead of costing the run its line.

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


def algorithmic_bytes(stats, scene, width, rows, covered_px):
    """Per-kernel algorithmic bytes of one frame.

    raster_shade: 32 B per surviving triangle (its raster record) + 8 B per pixel (depth and colour,
        written once) + the texel lower bound min(16 B per covered pixel, all texture bytes).
    setup (vertex stage and triangle setup in one launch), counted for what this build moves: every
        mesh once (shared by its instances), one 64-B matrix per object, per surviving triangle its
        two records (32 + 48 B), 4 B per bin entry.  The per-unit figures for an engine that stores
        one vertex copy per object and passes clip positions through memory (48 B per instance
        vertex + 60 B per submitted triangle + 32 B per survivor) are reported beside it as
        `expanded_buffers_bytes`: they describe traffic this build does not have.
    """
    t, tv, vi = stats["prims"], stats["records"], stats["instance_vertices"]
    tex_bytes = sum(int(x.nbytes) for x in scene.textures)
    texel = min(covered_px * 16, tex_bytes)
    mesh_bytes = int(scene.vertices.nbytes) + int(scene.indices.nbytes)
    return {
        "setup": mesh_bytes + stats["objects"] * 64 + tv * (32 + 48) + stats["bin_entries"] * 4,
        "setup_expanded_buffers": vi * 48 + t * 60 + tv * 32,
        "raster_shade": tv * 32 + width * rows * 8 + texel,
    }


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(scene, width, height, n_prims):
    """The CPU oracle (band-parallel variant): rotation update + one full frame per step of the
    same workload, on the host cores this process may use."""
    from oracle import oracle
    oracle.build()
    # the box reports every CPU of the machine; OMP_NUM_THREADS carries this job's share of them
    share = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    cores = min(os.cpu_count() or 1, share) if share > 0 else (os.cpu_count() or 1)
    cores = max(1, min(cores, 256))
    view, proj = oracle.camera(scene.camera, width, height)
    transforms = scene.transforms.copy()
    t0 = time.perf_counter()
    frames = 0
    while True:
        oracle.rotation_system(transforms, SEED, frames, DELTA_MS)
        oracle.render(scene, oracle.to_model(transforms), view, proj, width, height, threads=cores)
        frames += 1
        dt = time.perf_counter() - t0
        if dt > 12.0 or frames >= 64:
            break
    # config 1 (the reference's own CPU-runnable case: one cube, 800x600) on one thread
    from renderer_amd import scenes as _scenes
    cube = _scenes.config_scene(1)
    cw, ch = cube.size
    cview, cproj = oracle.camera(cube.camera, cw, ch)
    cmodels = oracle.to_model(cube.transforms)
    oracle.render(cube, cmodels, cview, cproj, cw, ch, threads=1)
    t1 = time.perf_counter()
    for _ in range(5):
        oracle.render(cube, cmodels, cview, cproj, cw, ch, threads=1)
    cube_ms = 1e3 * (time.perf_counter() - t1) / 5.0
    return {"value": n_prims * frames / dt, "unit": "triangles/s", "cores": cores, "kind": "port",
            "config1_single_thread_ms_per_frame": cube_ms,
            "cpu": cpu_model(), "cpus_visible": os.cpu_count(),
            "sample": "%d step(s) of the same workload (rotation update, model matrices, one full frame each), %.1f s "
                      "of CPU work, band-parallel oracle with %d threads (this job's share of the host: "
                      "OMP_NUM_THREADS)" % (frames, dt, cores),
            "ms_per_frame": 1e3 * dt / frames}


def profile_file(name):
    path = os.path.join(ROOT, "profiles", name)
    try:
        return json.load(open(path))
    except (OSError, ValueError):
        return None


def measure_frame_parallel(api, torch, dist, scene, local_rank, world, rank, steps, warmup, prewarm, stat_device):
    """The frame-parallel split of N ranks: rank r draws animation frames r, r + N, ... (every rank
    advances the replicated rotation state through all frames), no exchange.  One step = N frames.
    Exactly `steps` steps between fences, the maximum over the ranks."""
    width, height = scene.size
    d2 = api.HipDrawer(width, height, device=local_rank, n_images=2)
    d2.update_scene(scene)
    d2.update_camera(scene.camera)
    done = 0  # frames of the animation whose rotation update this rank has applied

    def frame_step(i):
        nonlocal done
        target = i * world + rank
        if done <= target:
            d2.rotation_system(SEED, done, DELTA_MS, n_frames=target - done + 1)
            done = target + 1
        d2.update_camera(scene.camera)
        d2.draw(i & 1)

    def fence():
        d2.sync()
        torch.cuda.synchronize()
        dist.barrier()

    for i in range(prewarm + warmup):
        frame_step(i)
    fence()
    t0 = time.perf_counter()
    for i in range(steps):
        frame_step(prewarm + warmup + i)
    fence()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=stat_device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    d2.close()
    elapsed = float(t.item())
    return {"value": scene.n_triangles * steps * world / elapsed, "unit": "triangles/s", "scaling": "weak",
            "steps": steps, "warmup": warmup, "frames_per_step": world, "ms_per_step": 1e3 * elapsed / max(steps, 1),
            "frames_per_sec": steps * world / elapsed, "parallelism": "frame-parallel%d, no exchange" % world}


def fallback_line(args, scene, world, fp, why):
    """The line of a sort-first run whose band exchange did not come through: the frame-parallel
    measurement taken before it, under the same contract, and what happened to the other."""
    width, height = scene.size
    line = {"metric": "triangles_per_sec", "value": fp["value"], "unit": "triangles/s", "n_gpus": world,
            "steps": fp["steps"], "warmup": fp["warmup"], "ms_per_step": fp["ms_per_step"], "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32+i64", "data": "synthetic",
            "frames_per_sec": fp["frames_per_sec"],
            "config": {"workload": "config %d: %s, %d objects x 1024 triangles = %d triangles, %dx%d, per-frame device-side "
                                   "rotation update" % (args.config, scene.name, len(scene.objects), scene.n_triangles, width, height),
                       "parallelism": "frame-parallel%d" % world, "frames_per_step": world},
            "roofline": None, "cpu_baseline": None,
            "sortfirst": {"error": why, "note": "the sort-first measurement did not finish; this line is the frame-parallel "
                                                "split of the same workload, measured before it"}}
    if args.scale != 1.0:
        line["invalid"] = "scaled-down debug run"
    if os.environ.get("SXE_BENCH_REHEARSAL", "") == "1":
        line["invalid"] = "rehearsal: ranks share GPUs, exchange over gloo through host memory"
    return line


class Watchdog:
    """Ends a multi-rank run whose exchange hangs: after `seconds` without finish(), rank 0 prints
    the line it was given and every rank leaves (a rank stuck inside a collective cannot be
    interrupted from Python, so the process ends from this thread)."""

    def __init__(self, seconds, rank):
        import threading
        self.line, self.rank, self.seconds = None, rank, seconds
        self.done = threading.Event()
        self.thread = threading.Thread(target=self.run, daemon=True)

    def start(self, line, key="sortfirst"):
        self.line, self.key = line, key
        if self.seconds > 0:
            self.thread.start()

    def run(self):
        if not self.done.wait(self.seconds):
            self.leave("no result within %d s" % self.seconds)

    def leave(self, why):
        if self.rank == 0 and self.line is not None:
            self.line.setdefault(self.key, {})["error"] = why
            print(json.dumps(self.line), flush=True)
        sys.stderr.write("[bench] rank %d leaves: sort-first measurement failed (%s)\n" % (self.rank, why))
        sys.stderr.flush()
        os._exit(0 if self.line is not None or self.rank != 0 else 3)

    def finish(self):
        self.done.set()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mode", choices=["auto", "sortfirst", "frames"], default="auto")
    ap.add_argument("--config", type=int, default=3, choices=[3, 4])
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the workload (debug only; marks the line invalid)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-depth-variant", action="store_true",
                    help="leave out the further measurements (depth images not kept; two frames in flight): kernel traces of the headline alone")
    ap.add_argument("--prewarm-frames", type=int, default=2048,
                    help="untimed steps before the warm-up steps, so that the GPU's clocks have ramped up (0: none)")
    ap.add_argument("--check", action="store_true", help="compare the last frame with the oracle (slow)")
    ap.add_argument("--bands", choices=["auto", "even"], default="auto",
                    help="sortfirst: band heights evened out against the measured gather (auto) or equal")
    ap.add_argument("--pipeline", action="store_true",
                    help="with one GPU: go through the sort-first gather pipeline anyway (its band is the whole frame)")
    ap.add_argument("--watchdog-seconds", type=int, default=240,
                    help="N > 1, sort-first: time allowed for that measurement before the run ends with the "
                         "frame-parallel line taken first (0: no limit)")
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
    # SXE_BENCH_REHEARSAL=1: a dry run of the N-rank path where there are fewer GPUs than ranks (the
    # ranks share the GPUs, gloo carries the exchange through host memory); its line is marked invalid
    rehearsal = os.environ.get("SXE_BENCH_REHEARSAL", "") == "1"
    if rehearsal:
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    stat_device = "cpu" if rehearsal else "cuda"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    scene = scenes.config_scene(args.config, scale=args.scale)
    width, height = scene.size
    n_prims = scene.n_triangles
    mode = args.mode
    if mode == "auto":
        mode = "sortfirst" if args.config == 4 else "frames"
    sortfirst = (mode == "sortfirst" and world > 1) or (world == 1 and args.pipeline)
    band = shard.band_rows(height, world, rank)
    if world > 1 and band[1] <= band[0] and (sortfirst or not args.no_depth_variant):
        sys.exit("rank %d has no tile row to render (%d ranks for %d rows)" % (rank, world, height))

    def make_drawer(scissor):
        d = api.HipDrawer(width, height, device=local_rank, n_images=2, scissor=scissor, timing=True)
        d.update_scene(scene)
        d.update_camera(scene.camera)
        return d

    def measure(sort_first):
        return run_measurement(args, api, torch, dist, shard, scene, world, rank, local_rank, rehearsal, stat_device, sort_first,
                               band if sort_first else None, make_drawer)

    def guarded(watchdog, what):
        # whatever the exchange raises, the line the watchdog holds still goes out
        try:
            return what()
        except Exception as e:  # noqa: BLE001
            import traceback
            traceback.print_exc()
            watchdog.leave("%s: %s" % (type(e).__name__, str(e)[:300]))

    if world == 1:
        result = measure(sortfirst)
    elif sortfirst:
        # Sort-first is the headline: the other split first (every rank draws whole frames of its own,
        # no exchange).  It is reported beside the headline, and it is the line of a run whose band
        # exchange fails or hangs.
        frame_parallel = measure_frame_parallel(api, torch, dist, scene, local_rank, world, rank, args.steps, args.warmup,
                                                min(max(0, args.prewarm_frames), 512), stat_device)
        watchdog = Watchdog(args.watchdog_seconds, rank)
        watchdog.start(fallback_line(args, scene, world, frame_parallel, "") if rank == 0 else None, "sortfirst")
        result = guarded(watchdog, lambda: measure(True))
        watchdog.finish()
        if rank == 0:
            result["frame_parallel"] = frame_parallel  # (measured before the headline, on a context of its own)
    else:
        # Frame-parallel is the headline; the sort-first split of the same frames as a further key.
        result = measure(False)
        if not args.no_depth_variant:
            watchdog = Watchdog(args.watchdog_seconds, rank)
            watchdog.start(result if rank == 0 else None, "sort_first")
            sf = guarded(watchdog, lambda: measure(True))
            watchdog.finish()
            if rank == 0:
                keep = ("value", "unit", "steps", "warmup", "ms_per_step", "scaling", "frames_per_sec", "gather_ms", "bands",
                        "band_tuning", "per_rank", "stale_bands_in_timed_region", "kernels_ms", "sustained", "check", "invalid")
                result["sort_first"] = {k: sf[k] for k in keep if k in sf}
                result["sort_first"]["parallelism"] = sf["config"]["parallelism"]
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


def run_measurement(args, api, torch, dist, shard, scene, world, rank, local_rank, rehearsal, stat_device, sortfirst,
                    scissor, make_drawer):
    """One split of the workload over the ranks, measured under the contract (warm-up, exactly
    --steps steps between fences and barriers, the maximum over the ranks): returns the line's
    fields on rank 0, None elsewhere."""
    width, height = scene.size
    n_prims = scene.n_triangles
    if os.environ.get("SXE_BENCH_TEST_FAIL") == "raise" and sortfirst and world > 1 and rank == world - 1:
        raise RuntimeError("injected failure (test of the fallback line)")
    if os.environ.get("SXE_BENCH_TEST_FAIL") == "hang" and sortfirst and world > 1 and rank == world - 1:
        time.sleep(10 ** 6)
    drawer = make_drawer(scissor)
    weights, tuning, first_frame = None, None, 0
    if sortfirst and world > 1 and args.bands == "auto":
        # Equal bands first: what a frame costs this rank apart from its rows (geometry stage,
        # replicated), what its rows cost, and what the gather of equal bands costs on its own.
        for i in range(12):
            if i == 4:
                drawer.sync()
                drawer.reset_timing_totals()
            drawer.rotation_system(SEED, i, DELTA_MS)
            drawer.draw(i & 1)
        drawer.sync()
        acc, frames = drawer.timing_totals()
        share = (scissor[1] - scissor[0]) / float(height)
        fixed_ms = (acc["object_ms"] + acc["setup_ms"] + acc["finish_ms"]) / max(frames, 1)
        band_ms = acc["raster_shade_ms"] / max(frames, 1) / share
        band = shard.band_tensor(drawer, 0, scissor[0], scissor[1])
        landing = torch.empty((height, width, 4), dtype=torch.uint8, device=band.device) if rank == 0 else None
        for k in range(8):
            if k == 3:
                torch.cuda.synchronize()
                dist.barrier()
                t_g = time.perf_counter()
            shard.gather_bands(band, height, width, world, rank, out=landing)
        torch.cuda.synchronize()
        dist.barrier()
        link_ms = 1e3 * (time.perf_counter() - t_g) / 5.0 * world  # (a whole frame over one link)
        del band, landing
        agreed = torch.tensor([fixed_ms, band_ms, link_ms], dtype=torch.float64, device=stat_device)
        dist.all_reduce(agreed, op=dist.ReduceOp.MAX)
        fixed_ms, band_ms, link_ms = [float(x) for x in agreed.tolist()]
        weights = shard.balanced_weights(world, 0, fixed_ms, band_ms, link_ms)
        tuning = {"fixed_ms": fixed_ms, "band_ms_per_frame_of_rows": band_ms, "gather_ms_per_frame_over_one_link": link_ms,
                  "weights": weights}
        if len(set(weights)) == 1:
            weights = None
        new_band = shard.band_rows(height, world, rank, weights=weights)
        if new_band != scissor:
            drawer.close()
            scissor = new_band
            drawer = make_drawer(scissor)
        else:
            first_frame = 12  # (the rotation system goes on where the trial frames left it)
    # (strict=False: frames drawn again after their bands had left are counted, and reported for the timed region)
    pipe = shard.SortFirstPipeline(drawer, height, width, world, rank, weights=weights, strict=False) if sortfirst else None

    frame_no = [first_frame]

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
        if pipe:
            pipe.submit(i)  # draw into image i & 1, gather on a second stream; no host wait
        else:
            drawer.draw(i & 1)

    def fence():
        if pipe:
            pipe.finish()
        drawer.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    # Clocks first: a GPU that has been idle runs its first tens of milliseconds below its sustained
    # clocks (the same 20 steps take 7 % longer after 5 frames of load than after 500), so the same
    # step runs untimed --prewarm-frames times before the --warmup steps.  Neither is in the timed region.
    prewarm_frames = max(0, args.prewarm_frames)  # (a count, not a time: every rank must take the same number of steps)
    for i in range(prewarm_frames):
        step(i)
    fence()
    for i in range(args.warmup):
        step(prewarm_frames + i)
    fence()
    redraws_before = drawer.stats()["redraws"]
    stale_before = pipe.stale_frames if pipe else 0
    # The headline: exactly --steps steps between two fences, without any timing event on the
    # stream (every event costs a few microseconds of stream time, six of them per frame).
    drawer.set_timing_stride(0)
    t0 = time.perf_counter()
    for i in range(args.steps):
        # frames are enqueued back to back (the library keeps up to three in flight on its
        # stream); nothing waits on the host until the closing fence
        step(prewarm_frames + args.warmup + i)
    fence()
    elapsed = time.perf_counter() - t0
    redraws_timed = drawer.stats()["redraws"] - redraws_before
    stale_timed = (pipe.stale_frames - stale_before) if pipe else 0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=stat_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    next_step = prewarm_frames + args.warmup + args.steps
    # Per-kernel device times come from a pass of their own after the headline: the same step, every
    # kernel of every frame bracketed by HIP events the library records on its own stream.
    sampled_frames = 16
    drawer.set_timing_stride(1)
    drawer.reset_timing_totals()
    for i in range(sampled_frames):
        step(next_step + i)
    fence()
    next_step += sampled_frames
    acc, timed_frames = drawer.timing_totals()
    stats = drawer.stats()
    kern = {k: v / max(timed_frames, 1) for k, v in acc.items()}
    # Sustained rate: the same step for at least half a second (again without events), so that the
    # closing fence is spread over thousands of frames and a sampler on the outside sees the GPU busy.
    drawer.set_timing_stride(0)
    sustained_steps = int(min(max(0.6 * args.steps / max(elapsed, 1e-6), 32), 20000))
    t0 = time.perf_counter()
    for i in range(sustained_steps):
        step(next_step + i)
    fence()
    sustained_elapsed = time.perf_counter() - t0
    next_step += sustained_steps
    last_step = next_step - 1
    gather_ms = None
    if pipe:
        gather_ms = 0.5 * (pipe.last_gather_ms(last_step) + pipe.last_gather_ms(last_step - 1))
    per_rank = None
    if world > 1:
        t = torch.tensor([sustained_elapsed], dtype=torch.float64, device=stat_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        sustained_elapsed = float(t.item())
        mine = torch.tensor([kern["object_ms"], kern["setup_ms"], kern["finish_ms"], kern["raster_shade_ms"], kern["frame_ms"],
                             gather_ms or 0.0, float(stats["records"]), float(stats["bin_entries"]),
                             float(scissor[1] - scissor[0]) if scissor else float(height)], dtype=torch.float64, device=stat_device)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        names = ["object_ms", "setup_ms", "finish_ms", "raster_shade_ms", "frame_ms", "gather_ms", "records", "bin_entries", "rows"]
        per_rank = [dict(zip(names, [float(x) for x in r.tolist()])) for r in allr]

    frames_per_step = 1 if (sortfirst or world == 1) else world
    total_frames = frames_per_step * args.steps
    ms_per_step = 1e3 * elapsed / max(args.steps, 1)
    tri_per_s = n_prims * total_frames / elapsed
    fps = total_frames / elapsed

    result = None
    if rank == 0:
        img = last_step & 1
        rows = (scissor[1] - scissor[0]) if scissor else height
        depth = drawer.read_depth(img)
        covered_px = int((depth[scissor[0]:scissor[1]] < 1.0).sum()) if scissor else int((depth < 1.0).sum())
        ab = algorithmic_bytes(stats, scene, width, rows, covered_px)
        stage_ms = {"setup": kern["setup_ms"], "raster_shade": kern["raster_shade_ms"]}
        # the two large kernels take the same time to within run-to-run noise: the tile kernel is
        # reported as long as it is within 5 % of the longest, so that the line does not flip between runs
        dom = "raster_shade" if stage_ms["raster_shade"] >= 0.95 * stage_ms["setup"] else "setup"
        achieved = ab[dom] / (stage_ms[dom] * 1e-3) / 1e9 if stage_ms[dom] > 0 else 0.0
        traffic_all = profile_file("pmc_traffic.json") or {}
        valu_all = profile_file("pmc_valu_lds.json") or {}
        mem_all = profile_file("pmc_mem.json") or {}
        # the counter summaries are artefacts of tools/profile_round.sh, not of this run: they are stale
        # when the kernel sources have changed since they were collected
        from renderer_amd import _build
        built_from = _build.source_hash()
        counters_stale = any(f.get("source_hash") != built_from for f in (traffic_all, valu_all, mem_all))
        valu = next((v for k, v in valu_all.items() if isinstance(v, dict) and ("k_" + dom) in k), {})
        mem = next((v for k, v in mem_all.items() if isinstance(v, dict) and ("k_" + dom) in k), {})
        roofline = {"bound": "hbm", "kernel": "k_" + dom, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic_all.get(dom),
                    "traffic_fetch_factor": traffic_all.get("fetch_factor"),
                    # the bytes the kernel really moves at the L2's memory side (counters) over its
                    # duration: how close the memory system is to its limit whatever the byte model says
                    "traffic_GBps": (traffic_all.get(dom) / (stage_ms[dom] * 1e-3) / 1e9) if traffic_all.get(dom) and stage_ms[dom] > 0 else None,
                    "traffic_frac": (traffic_all.get(dom) / (stage_ms[dom] * 1e-3) / 1e9 / HBM_PEAK_GBPS) if traffic_all.get(dom) and stage_ms[dom] > 0 else None,
                    # (false: profiles/pmc_*.json were collected on exactly these kernel sources)
                    "counters_stale": counters_stale, "counters_source_hash": traffic_all.get("source_hash"),
                    "source_hash": built_from,
                    "algorithmic_bytes_per_launch": ab[dom], "avg_ms_per_launch": stage_ms[dom],
                    # what binds the kernel in practice (tools/pmc_valu_lds.sh, profiles/pmc_valu_lds.json):
                    # share of the launch during which the vector ALUs of all SIMDs were issuing
                    "valu_frac": valu.get("valu_busy_frac"), "valu_lane_use": valu.get("valu_lane_use"),
                    "lds_frac": valu.get("lds_busy_frac"),
                    # the vector-memory pipeline of the compute units (tools/pmc_mem.sh, profiles/pmc_mem.json):
                    # share of the launch the data-return units were busy, and the L1s were held up by
                    # requests still on their way to or from the L2
                    "td_busy_frac": mem.get("td_busy_frac"), "l1_pending_stall_frac": mem.get("tcp_pending_stall_frac"),
                    "l2_read_requests": mem.get("TCP_TCC_READ_REQ_sum")}
        assert roofline["frac"] <= 1.0, "algorithmic bytes over time exceed the HBM peak: the byte model is wrong"
        stage = {}
        for k in stage_ms:
            g = ab[k] / (stage_ms[k] * 1e-3) / 1e9 if stage_ms[k] > 0 else 0.0
            v = next((x for n, x in valu_all.items() if isinstance(x, dict) and ("k_" + k) in n), {})
            stage[k] = {"algorithmic_bytes": ab[k], "avg_ms": stage_ms[k], "achieved_GBps": g, "hbm_frac": g / HBM_PEAK_GBPS,
                        "traffic": traffic_all.get(k), "valu_frac": v.get("valu_busy_frac")}
        stage["setup"]["expanded_buffers_bytes"] = ab["setup_expanded_buffers"]
        cfg_name = "config %d: %s, %d objects x 1024 triangles = %d triangles, %dx%d, per-frame device-side rotation update" % (
            args.config, scene.name, len(scene.objects), n_prims, width, height)
        result = {
            "metric": "triangles_per_sec", "value": tri_per_s, "unit": "triangles/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong" if sortfirst else "weak", "vs_baseline": None, "dtype": "f32+i64",
            "data": "synthetic",
            "frames_per_sec": fps,
            "config": {"workload": cfg_name,
                       "parallelism": ("sortfirst-bands%d+rccl-gather-overlapped" % world) if sortfirst else
                                      ("frame-parallel%d" % world if world > 1 else "single"),
                       "frames_per_step": frames_per_step, "tile": stats["tile_size"]},
            "roofline": roofline,
            "stage_rooflines": stage,
            "kernels_ms": kern,
            "kernel_timing": {"pass": "separate pass of %d frames after the timed region, events around every kernel; the "
                                      "timed region itself records no events" % sampled_frames,
                              "frames_sampled": int(timed_frames), "events_in_timed_region": 0},
            "sustained": {"steps": sustained_steps, "seconds": sustained_elapsed,
                          "ms_per_step": 1e3 * sustained_elapsed / sustained_steps,
                          "value": n_prims * frames_per_step * sustained_steps / sustained_elapsed, "unit": "triangles/s"},
            "frame_stats": stats,
            "redraws_in_timed_region": redraws_timed,
            "prewarm": {"frames": prewarm_frames,
                        "why": "untimed frames before the --warmup steps so that the timed steps run at sustained clocks"},
        }
        if gather_ms is not None:
            result["gather_ms"] = gather_ms
            result["stale_bands_in_timed_region"] = stale_timed  # (frames drawn again after their bands had been gathered)
        if sortfirst:
            result["bands"] = shard.all_bands(height, world, weights=weights)
        if tuning is not None:
            result["band_tuning"] = tuning
        if per_rank is not None:
            result["per_rank"] = per_rank
        if args.scale != 1.0:
            result["invalid"] = "scaled-down debug run"
        if rehearsal:
            result["invalid"] = "rehearsal: ranks share GPUs, exchange over gloo through host memory"
        if world == 1 and not args.no_cpu_baseline:
            try:
                result["cpu_baseline"] = cpu_baseline(scene, width, height, n_prims)
            except Exception as e:  # noqa: BLE001 (a host without a compiler for the oracle still gets its GPU line)
                result["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        if args.check:
            from oracle import oracle
            t = drawer.read_transforms()
            ref = oracle.render(scene, oracle.to_model(t), *oracle.camera(scene.camera, width, height), width, height,
                                threads=16)
            got = pipe.frame.cpu().numpy() if pipe else drawer.read_color(img)
            y0, y1 = scissor if scissor else (0, height)  # (depth stays on the rank that drew it)
            result["check"] = {"color_equal": bool(np.array_equal(ref["color"], got)),
                               "depth_equal": bool(np.array_equal(ref["depth"][y0:y1].view(np.uint32), depth[y0:y1].view(np.uint32)))}
    drawer.close()
    def depth_variant():
        # The same frames with the depth images not kept, which is what the reference's pass does with its
        # depth attachment (AttachmentStoreOp::DontCare, src/renderer/drawer.rs:544-547).  A second key, not
        # the headline: the headline keeps the depth images, which the parity tests read.
        with api.HipDrawer(width, height, device=local_rank, n_images=2, discard_depth=True) as d3:
            d3.update_scene(scene)
            d3.update_camera(scene.camera)
            dd_steps = max(200, args.steps)
            for i in range(256):
                d3.rotation_system(SEED, i, DELTA_MS)
                d3.draw(i & 1)
            d3.sync()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(dd_steps):
                d3.rotation_system(SEED, 256 + i, DELTA_MS)
                d3.draw(i & 1)
            d3.sync()
            torch.cuda.synchronize()
            dd_elapsed = time.perf_counter() - t0
        return {"value": n_prims * dd_steps / dd_elapsed, "unit": "triangles/s", "steps": dd_steps,
                "ms_per_step": 1e3 * dd_elapsed / dd_steps,
                "what": "SXE_FLAG_DISCARD_DEPTH: the reference's store op for its depth attachment"}

    def overlap_variant():
        # The same steps with a context per target image (api.FramesInFlight), the way the reference
        # keeps a command buffer and a fence per swapchain image (src/renderer/drawer.rs:412-456): the
        # frames of the two images are on two streams and overlap on the GPU.  A further key, not the
        # headline: the kernel figures of the headline belong to frames that run one after the other.
        with api.FramesInFlight(width, height, n_images=2, device=local_rank) as d4:
            d4.update_scene(scene)
            d4.update_camera(scene.camera)
            d4.set_timing_stride(0)
            ov_steps = max(200, args.steps)
            for i in range(512):
                d4.rotation_system(SEED, i, DELTA_MS)
                d4.update_camera(scene.camera)
                d4.draw(i & 1)
            d4.sync()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(ov_steps):
                d4.rotation_system(SEED, 512 + i, DELTA_MS)
                d4.update_camera(scene.camera)
                d4.draw(i & 1)
            d4.sync()
            torch.cuda.synchronize()
            ov_elapsed = time.perf_counter() - t0
            ov_redraws = d4.stats()["redraws"]
        return {"value": n_prims * ov_steps / ov_elapsed, "unit": "triangles/s", "steps": ov_steps,
                "ms_per_step": 1e3 * ov_elapsed / ov_steps, "redraws": ov_redraws,
                "what": "a context (stream, work buffers) per target image: consecutive frames overlap "
                        "on the GPU; every frame is drawn, each bit for bit the frame of the headline"}

    if world == 1 and not sortfirst and not args.no_depth_variant:
        # (further keys: whatever goes wrong in them must not cost the run its line)
        for key, measure in (("depth_not_kept", depth_variant), ("two_frames_in_flight", overlap_variant)):
            try:
                result[key] = measure()
            except Exception as e:  # noqa: BLE001
                result[key] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
    pipe = None
    return result


if __name__ == "__main__":
    main()
