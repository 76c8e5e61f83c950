#!/usr/bin/env python3
"""Development: kernel times of a configuration under every shading mode (3 = the lit extension).

  lit_probe.py [config]
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderer_amd import api, scenes
s = scenes.config_scene(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
w, h = s.size
for shading in (0, 1, 2, 3):
    with api.HipDrawer(w, h, timing=True, shading=shading) as d:
        d.update_scene(s); d.update_camera(s.camera)
        for i in range(24):
            if i == 4:
                d.reset_timing_totals()
            d.rotation_system(3, i, 16.0); d.draw(i & 1); d.sync()
        sums, frames = d.timing_totals()
        print("shading %d" % shading, {k: round(v / frames, 4) for k, v in sums.items() if v})
