#!/usr/bin/env python3
"""Development helper: render a few frames of a benchmark configuration (for rocprofv3 runs).

  frames.py [config] [frames] [debug_flags]
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderer_amd import api, scenes
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dbg = int(sys.argv[3], 0) if len(sys.argv) > 3 else 0
s = scenes.config_scene(cfg)
w, h = s.size
with api.HipDrawer(w, h, timing=True, debug_flags=dbg) as d:
    d.update_scene(s); d.update_camera(s.camera)
    for i in range(n):
        if i == 2:
            d.reset_timing_totals()  # the first frames grow buffers and warm caches
        d.rotation_system(3, i, 16.0); d.draw(i & 1); d.sync()
    print(d.stats())
    sums, frames = d.timing_totals()
    print({k: round(v / max(frames, 1), 6) for k, v in sums.items()})  # mean over the frames after the second
