#!/usr/bin/env python3
"""Development probe: views per second of config 5 (one 512x512 view per frame) with the views dealt
out over several contexts, each with its own stream."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderer_amd import api, scenes
s = scenes.config_scene(5)
w, h = s.size
cams = scenes.orbit_cameras(512, 5)
for n_ctx in (1, 2, 4, 8):
    ds = [api.HipDrawer(w, h, n_images=2) for _ in range(n_ctx)]
    for d in ds:
        d.update_scene(s)
        for i in range(4):
            d.update_camera(cams[i:i + 1]); d.draw(i & 1)
        d.sync()
    t0 = time.perf_counter()
    n = 512
    for i in range(n):
        d = ds[i % n_ctx]
        d.update_camera(cams[i:i + 1]); d.draw((i // n_ctx) & 1)
    for d in ds:
        d.sync()
    dt = time.perf_counter() - t0
    print("contexts=%d: %.4f ms/view, %.0f views/s" % (n_ctx, 1e3 * dt / n, n / dt), flush=True)
    for d in ds:
        d.close()
