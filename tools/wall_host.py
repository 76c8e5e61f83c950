#!/usr/bin/env python3
"""Development probe: frame rate of config 3 when the host takes part in every frame the way the
reference's loop does (src/main.rs:189-197): model matrices uploaded per frame (update_uniforms),
optionally the finished frame read back (the Presenter stand-in)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from renderer_amd import api, scenes
s = scenes.config_scene(3)
w, h = s.size
models = api.to_model(s.transforms)
with api.HipDrawer(w, h, n_images=2) as d:
    d.update_scene(s); d.update_camera(s.camera)
    for mode in ("device-resident", "update_uniforms per frame", "update_uniforms + read_color per frame", "update_uniforms + map_color per frame"):
        for i in range(5):
            if mode == "device-resident":
                d.rotation_system(3, i, 16.0)
            d.draw(i & 1)
        d.sync()
        n = 100 if "color" not in mode else 30
        t0 = time.perf_counter()
        for i in range(n):
            if mode == "device-resident":
                d.rotation_system(3, 10 + i, 16.0)
            else:
                d.update_uniforms(models)
            d.update_camera(s.camera)
            d.draw(i & 1)
            if "read" in mode:
                d.read_color(i & 1)
            if "map" in mode:
                d.map_color(i & 1)
        d.sync()
        dt = (time.perf_counter() - t0) / n
        print("%-40s %.3f ms/frame  %.3e triangles/s" % (mode, 1e3 * dt, s.n_triangles / dt), flush=True)
