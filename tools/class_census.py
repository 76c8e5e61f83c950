import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from renderer_amd import api, scenes
s = scenes.config_scene(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
w, h = s.size
with api.HipDrawer(w, h) as d:
    d.update_scene(s); d.update_camera(s.camera)
    for i in range(2):
        d.rotation_system(3, i, 16.0); d.draw(0); d.sync()
    tp = d.tile_parts().astype(np.int64)
print("class sums", tp.sum(0), "tiles", len(tp))
n = tp[:, :4].sum(1)
print("mean entries/tile", n.mean(), "nonempty tiles", (n > 0).sum())
# wave-iteration model: entries sorted largest class first, 64 per wave; iterations = upper bound of largest class in wave
ub = [2, 8, 24, 64]
tot = 0
for c0, c1, c2, c3, m in tp:
    seq = [3] * int(c3 + m) + [2] * int(c2) + [1] * int(c1) + [0] * int(c0)
    for k in range(0, len(seq), 64):
        tot += ub[seq[k]]
print("model wave-iterations per tile", tot / len(tp))
