#!/usr/bin/env python3
"""Development helper: do frames of config 3 on two streams of one GPU overlap to any gain?

Two contexts (each with its own stream and work buffers) are fed alternately from one host thread;
the aggregate frame rate is compared with one context's.
"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from renderer_amd import api, scenes

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
s = scenes.config_scene(cfg)
w, h = s.size
ds = []
n_ctx = int(sys.argv[2]) if len(sys.argv) > 2 else 2
for k in range(n_ctx):
    d = api.HipDrawer(w, h, timing=False)
    d.update_scene(s); d.update_camera(s.camera); d.set_timing_stride(0)
    ds.append(d)


def run(drawers, frames, first):
    for i in range(frames):
        for d in drawers:
            d.rotation_system(3, first + i, 16.0); d.draw(i & 1)
    for d in drawers:
        d.sync()
    torch.cuda.synchronize()


for rounds in range(3):
    for name, group in [("%d context(s)" % k, ds[:k]) for k in range(1, n_ctx + 1)]:
        run(group, 300, 0)
        t0 = time.perf_counter()
        n = 1500
        run(group, n, 300)
        dt = time.perf_counter() - t0
        print("%s: %.4f ms per frame (%d frames in %.1f ms)" % (name, 1e3 * dt / (n * len(group)), n * len(group), 1e3 * dt))
for d in ds:
    d.close()
