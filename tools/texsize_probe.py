"""Development: config 3 under textures of other sizes (how much of the tile kernel is the size of the footprint tables)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderer_amd import api, scenes
for ts in (512, 256, 128, 64):
    s = scenes.sphere_field_scene(10000, 3, 60.0, (-200.0, -5.0), 3840, 2160, tex_size=ts, name="cfg3-tex%d" % ts)
    w, h = s.size
    with api.HipDrawer(w, h, timing=True) as d:
        d.update_scene(s); d.update_camera(s.camera)
        for i in range(100):
            if i == 4: d.reset_timing_totals()
            d.rotation_system(3, i, 16.0); d.draw(i & 1); d.sync()
        sums, frames = d.timing_totals()
        print("texture %d^2:" % ts, {k: round(v / max(frames, 1), 4) for k, v in sums.items()}, flush=True)
