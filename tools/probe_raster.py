#!/usr/bin/env python3
"""Development probe: splits k_raster_shade time into its raster and resolve halves and prints the
bin-occupancy distribution of config 3.  Run on the GPU box."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from renderer_amd import api, scenes

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
s = scenes.config_scene(cfg)
w, h = s.size
for name, dbg in (("full", 0), ("no-resolve", 1 << 31), ("no-resolve,no-small", (1 << 31) | (1 << 28)), ("no-resolve,no-large", (1 << 31) | (1 << 29)), ("no-resolve,no-small,no-large", (1 << 31) | (3 << 28)), ("no-resolve,no-atomic", (1 << 31) | (1 << 27))):
    with api.HipDrawer(w, h, timing=True, debug_flags=dbg) as d:
        d.update_scene(s)
        d.update_camera(s.camera)
        acc = {}
        for i in range(12):
            d.rotation_system(3, i, 16.0)
            d.draw(i & 1)
            d.sync()
            if i >= 2:
                for k, v in d.timings().items():
                    acc[k] = acc.get(k, 0) + v / 10
        print(name, {k: round(v * 1000, 1) for k, v in acc.items()})
        if dbg == 0:
            tc = d.tile_counts()
            st = d.stats()
            print("stats", st)
            print("tiles", len(tc), "nonempty", int((tc > 0).sum()), "mean", tc.mean(), "max", tc.max(),
                  "p50/p90/p99/p999", np.percentile(tc, [50, 90, 99, 99.9]).tolist())
            order = np.sort(tc)[::-1]
            print("top16", order[:16].tolist(), "sum top 256:", int(order[:256].sum()), "of", int(tc.sum()))
