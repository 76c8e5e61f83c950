#!/usr/bin/env python3
"""Development helper: what a timed region of K frames of config 3 costs beyond K frames.

Regions of several lengths between the fences bench.py uses (the library's sync, then
torch.cuda.synchronize), the fence alone on an idle GPU, and the host's share of one step.
"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from renderer_amd import api, scenes

s = scenes.config_scene(3)
w, h = s.size
with api.HipDrawer(w, h, timing=False) as d:
    d.update_scene(s); d.update_camera(s.camera)
    frame = [0]

    def step():
        d.rotation_system(3, frame[0], 16.0); d.update_camera(s.camera); d.draw(frame[0] & 1)
        frame[0] += 1

    def fence():
        d.sync(); torch.cuda.synchronize()

    d.set_timing_stride(0)
    for _ in range(600):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(200):
        fence()
    print("fence on an idle GPU: %.1f us" % (1e6 * (time.perf_counter() - t0) / 200))
    for k in (1, 2, 3, 5, 10, 20, 50, 100, 400):
        best = []
        for _ in range(7):
            t0 = time.perf_counter()
            for _ in range(k):
                step()
            t1 = time.perf_counter()
            fence()
            t2 = time.perf_counter()
            best.append((t2 - t0, t1 - t0))
        best.sort()
        tot, host = best[len(best) // 2]
        print("K=%3d: region %.1f us, %.2f us per frame, enqueue loop %.1f us (%.1f per step)" % (
            k, 1e6 * tot, 1e6 * tot / k, 1e6 * host, 1e6 * host / k))
