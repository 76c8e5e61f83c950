#!/usr/bin/env python3
"""Development helper: what the rotation updates of the frames other ranks draw cost a rank.

A frame-parallel rank of N applies N updates per frame it draws (one launch, k_object).  Prints
k_object's and the frame's time on config 3 for spans of 1, 2, 4 and 8 updates per frame.
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderer_amd import api, scenes

s = scenes.config_scene(3)
w, h = s.size
with api.HipDrawer(w, h, timing=True) as d:
    d.update_scene(s); d.update_camera(s.camera)
    frame = 0
    for span in (1, 2, 4, 8, 1):
        for i in range(40):
            if i == 8:
                d.sync(); d.reset_timing_totals()
            d.rotation_system(3, frame, 16.0, n_frames=span); frame += span
            d.draw(i & 1)
        d.sync()
        sums, n = d.timing_totals()
        print("updates per frame %d: k_object %.2f us, frame %.2f us" % (span, 1e3 * sums["object_ms"] / n, 1e3 * sums["frame_ms"] / n))
