#!/usr/bin/env python3
"""Development: where the workgroups of the ordered (blending) tile kernel spend their time, on config 2
with every texture translucent."""
import sys, os
import numpy as np
os.environ["SXE_DEVELOPMENT"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderer_amd import api, scenes
s = scenes.config_scene(2)
s.textures = [scenes.translucent_texture(512, 100 + k) for k in range(4)]
w, h = s.size
with api.HipDrawer(w, h, debug_flags=1 << 26) as d:
    d.update_scene(s); d.update_camera(s.camera)
    for i in range(4):
        d.draw(i & 1); d.sync()
    t = d.tile_times().astype(np.int64)
    parts = d.tile_parts()
    blend = d.tile_counts() > 0  # (every texture is translucent: a tile with an entry takes the ordered kernel)
    t0 = t[blend, 0].min()
    order = [k for k in np.argsort(-(t[:, 3] - t[:, 0])) if blend[k]]
    print("kernel span %.1f us over %d tiles" % ((t[blend, 3].max() - t0) / 100.0, int(blend.sum())))
    print("tile: entries [by size class, late]; start, gather, sort, entries, out  [us]")
    for k in order[:12]:
        print(k, t[k, 5], parts[k].tolist(), (t[k, 0] - t0) / 100.0, (t[k, 4] - t[k, 0]) / 100.0, (t[k, 1] - t[k, 4]) / 100.0,
              (t[k, 2] - t[k, 1]) / 100.0, (t[k, 3] - t[k, 2]) / 100.0)
