import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, os.getcwd())
from renderer_amd import api, scenes
s = scenes.config_scene(3)
w, h = s.size
with api.HipDrawer(w, h, timing=True, debug_flags=(1 << 27)) as d:
    d.update_scene(s); d.update_camera(s.camera)
    d.rotation_system(3, 0, 16.0); d.draw(0); d.sync()
    print(d.stats())
