#!/usr/bin/env python3
"""Development: n frames of config 2 with every texture translucent (the ordered path), for profilers.

  blend_frames.py [frames]
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderer_amd import api, scenes
s = scenes.config_scene(2)
s.textures = [scenes.translucent_texture(512, 100 + k) for k in range(4)]
w, h = s.size
with api.HipDrawer(w, h) as d:
    d.update_scene(s); d.update_camera(s.camera)
    for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
        d.draw(i & 1); d.sync()
