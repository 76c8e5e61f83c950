"""Development: kernel times of configs 3 and 4 with the depth images kept and not kept (SXE_FLAG_DISCARD_DEPTH)."""
import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderer_amd import api, scenes
for cfg in (3,4):
    s=scenes.config_scene(cfg); w,h=s.size
    for dd in (False,True,False,True):
        with api.HipDrawer(w,h,timing=True,discard_depth=dd) as d:
            d.update_scene(s); d.update_camera(s.camera)
            for i in range(150):
                if i==10: d.reset_timing_totals()
                d.rotation_system(3,i,16.0); d.draw(i&1); d.sync()
            sums,frames=d.timing_totals()
            print("cfg",cfg,"discard_depth",dd,{k:round(v/max(frames,1),4) for k,v in sums.items()},flush=True)
