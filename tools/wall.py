#!/usr/bin/env python3
"""Development probe: wall clock per frame of config 3 with and without per-kernel events."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderer_amd import api, scenes
s = scenes.config_scene(3)
w, h = s.size
for timing in (False, True, False, True):
    with api.HipDrawer(w, h, n_images=2, timing=timing) as d:
        d.update_scene(s); d.update_camera(s.camera)
        for i in range(10):
            d.rotation_system(3, i, 16.0); d.draw(i & 1)
        d.sync()
        t0 = time.perf_counter()
        n = 200
        for i in range(n):
            d.rotation_system(3, 10 + i, 16.0); d.update_camera(s.camera); d.draw(i & 1)
        d.sync()
        print("timing=%s: %.4f ms/frame" % (timing, 1e3 * (time.perf_counter() - t0) / n))
