#!/usr/bin/env python3
"""Cost of the ordered (blending) path: config 2 (100 spheres, 1920x1080) with opaque textures, with
one of four textures translucent, and with all of them translucent."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderer_amd import api, scenes
s = scenes.config_scene(2)
w, h = s.size
variants = {"opaque": list(s.textures),
            "one of four translucent": [scenes.translucent_texture(512, 100)] + list(s.textures[1:]),
            "all translucent": [scenes.translucent_texture(512, 100 + k) for k in range(4)]}
for name, texs in variants.items():
    s.textures = texs
    with api.HipDrawer(w, h, timing=True) as d:
        d.update_scene(s); d.update_camera(s.camera)
        for i in range(12):
            if i == 2:
                d.reset_timing_totals()
            d.draw(i & 1); d.sync()
        sums, frames = d.timing_totals()
        print("%-26s" % name, {k: round(v / frames, 4) for k, v in sums.items() if v})
        c = np.sort(d.tile_counts())[::-1]
        print("    entries per tile: largest", c[:6].tolist(), "mean %.0f, tiles %d, above 256: %d" % (c.mean(), c.size, int((c > 256).sum())))
