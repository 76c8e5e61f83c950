#!/usr/bin/env python3
"""Config 5 on one GPU: 512x512 views of the 100-sphere scene per second, one view per draw
(update_camera + draw on one stream) against sets of views that share their launches
(sxe_draw_views).  No readback inside the timed loops; frames stay in flight.

  views_probe.py [views]      (default 4096)
"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from renderer_amd import api, scenes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
s = scenes.config_scene(5)
w, h = s.size
cams = scenes.orbit_cameras(n, 5, radius=60.0, target=(0.0, 0.0, -22.5))
n_img = 256
with api.HipDrawer(w, h, n_images=n_img, timing=True) as d:
    d.update_scene(s)
    # one by one
    for rep in range(2):
        t0 = time.perf_counter()
        for v in range(min(n, 1024)):
            d.update_camera(cams[v:v + 1]); d.draw(v % n_img)
        d.sync()
        dt = time.perf_counter() - t0
    single = dt / min(n, 1024)
    print("one view per draw: %.4f ms/view, %.0f views/s" % (1e3 * single, 1.0 / single))
    d.update_camera(cams[5:6]); d.draw(5)
    ref = d.read_color(5)
    # sets
    for rep in range(2):
        d.reset_timing_totals()
        t0 = time.perf_counter()
        for v0 in range(0, n, n_img):
            d.draw_views(cams[v0:v0 + n_img], 0)
        d.sync()
        dt = time.perf_counter() - t0
    sums, frames = d.timing_totals()
    print("sets of views:     %.4f ms/view, %.0f views/s (%d views in %.2f ms; %d launch sets)" % (1e3 * dt / n, n / dt, n, 1e3 * dt, frames))
    print("per launch set (ms):", {k: round(v / max(frames, 1), 4) for k, v in sums.items()})
    d.draw_views(cams[:n_img], 0); d.sync()
    print("view 5 equal to the one drawn alone:", bool(np.array_equal(d.read_color(5), ref)), "| speed-up %.1fx" % (single / (dt / n)))
