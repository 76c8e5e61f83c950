import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from renderer_amd import api, scenes
for cfg in (3, 4, 2):
    s = scenes.config_scene(cfg)
    w, h = s.size
    with api.HipDrawer(w, h) as d:
        d.update_scene(s); d.update_camera(s.camera)
        for i in range(2):
            d.rotation_system(3, i, 16.0); d.draw(0); d.sync()
        tp = d.tile_parts().astype(np.int64)
    print("cfg", cfg, "tiles", len(tp), "class sums", tp.sum(0), "max per class", tp.max(0), "p99", np.percentile(tp, 99, axis=0), "max total", tp.sum(1).max())
