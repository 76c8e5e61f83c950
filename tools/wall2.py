#!/usr/bin/env python3
"""Development probe: aggregate frame rate of config 3 with one, two and three contexts (own streams) fed in turn."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderer_amd import api, scenes
s = scenes.config_scene(3)
w, h = s.size
for n_ctx in (1, 2, 3, 1, 2):
    ds = [api.HipDrawer(w, h, n_images=2, timing=False) for _ in range(n_ctx)]
    for d in ds:
        d.update_scene(s); d.update_camera(s.camera)
        for i in range(6):
            d.rotation_system(3, i, 16.0); d.draw(i & 1)
    for d in ds:
        d.sync()
    t0 = time.perf_counter()
    n = 240
    for i in range(n):
        d = ds[i % n_ctx]
        d.rotation_system(3, 10 + i, 16.0); d.update_camera(s.camera); d.draw((i // n_ctx) & 1)
    for d in ds:
        d.sync()
    print("contexts=%d: %.4f ms/frame" % (n_ctx, 1e3 * (time.perf_counter() - t0) / n), flush=True)
    for d in ds:
        d.close()
