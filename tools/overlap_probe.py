#!/usr/bin/env python3
"""Development: what would overlapping consecutive frames on two streams gain?  Two contexts (each
has its own stream and work buffers) draw the same scene alternately from one host thread.

  overlap_probe.py [config] [frames]
"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderer_amd import api, scenes
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
s = scenes.config_scene(cfg)
w, h = s.size
ds = [api.HipDrawer(w, h) for _ in range(2)]
for d in ds:
    d.update_scene(s); d.update_camera(s.camera)
    for i in range(4):
        d.rotation_system(3, i, 16.0); d.draw(i & 1)
    d.sync()

def run(drawers, frames):
    t0 = time.perf_counter()
    for i in range(frames):
        d = drawers[i % len(drawers)]
        d.rotation_system(3, i, 16.0); d.draw((i // len(drawers)) & 1)
    for d in drawers:
        d.sync()
    return (time.perf_counter() - t0) / frames * 1e3

for rep in range(3):
    one = run(ds[:1], n)
    two = run(ds, n)
    print("ms per frame: one context %.4f, two contexts alternating %.4f (x%.3f)" % (one, two, one / two))
for d in ds:
    d.close()
