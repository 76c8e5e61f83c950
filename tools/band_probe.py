#!/usr/bin/env python3
"""Development: kernel times of one rank's band of a sort-first split (config 3 or 4) on one GPU.

  band_probe.py [config] [world] [rank]
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderer_amd import api, scenes, shard
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ranks = [int(sys.argv[3])] if len(sys.argv) > 3 else range(world)
s = scenes.config_scene(cfg)
w, h = s.size
for rank in ranks:
    band = shard.band_rows(h, world, rank)
    with api.HipDrawer(w, h, timing=True, scissor=band) as d:
        d.update_scene(s); d.update_camera(s.camera)
        for i in range(40):
            if i == 4:
                d.reset_timing_totals()
            d.rotation_system(3, i, 16.0); d.draw(i & 1); d.sync()
        sums, frames = d.timing_totals()
        st = d.stats()
        print("rank %d/%d rows %s records %d:" % (rank, world, band, st["records"]), {k: round(v / max(frames, 1), 4) for k, v in sums.items()})
