#!/usr/bin/env python3
"""Development probe: per-tile workgroup phase clocks of k_raster_shade.

  timeline.py [config] [world rank]     (world, rank: only that rank's band of a sort-first split)
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from renderer_amd import api, scenes, shard
s = scenes.config_scene(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
w, h = s.size
band = shard.band_rows(h, int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else None
with api.HipDrawer(w, h, timing=True, debug_flags=(1 << 26), scissor=band) as d:
    d.update_scene(s); d.update_camera(s.camera)
    for i in range(3):
        d.rotation_system(3, i, 16.0); d.draw(0); d.sync()
    tt = d.tile_times().astype(np.int64); tc = d.tile_counts().astype(np.int64); tp = d.tile_parts().astype(np.float64)
    print(d.timings())
t0 = tt[:, 0].min()
st = (tt[:, 0] - t0) / 100.0; en = (tt[:, 7] - t0) / 100.0; du = en - st   # microseconds
print("kernel span %.1f us; tiles %d; mean dur %.1f, p50 %.1f p90 %.1f p99 %.1f max %.1f" % (en.max(), len(du), du.mean(), *np.percentile(du, [50, 90, 99]), du.max()))
print("sum of durations / span = avg concurrency %.0f workgroups" % (du.sum() / en.max()))
names = ["init", "loop(tid0)", "loop barrier", "queue phase", "compaction", "shade"]
idx = [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 7)]
for lo, hi in [(0, 1), (1, 100), (100, 400), (400, 1000), (1000, 2000), (2000, 99999)]:
    m = (tc >= lo) & (tc < hi)
    if m.any():
        print("entries [%d,%d): %d tiles, mean dur %.1f us, mean start %.1f, max end %.1f" % (lo, hi, m.sum(), du[m].mean(), st[m].mean(), en[m].max()))
        print("    phases: " + ", ".join("%s %.2f" % (n, ((tt[m, b] - tt[m, a]) / 100.0).mean()) for n, (a, b) in zip(names, idx)))
edges = np.linspace(0, en.max(), 11)
print("running workgroups at deciles of the span:", [int(((st <= t) & (en > t)).sum()) for t in edges[:-1] + 0.5 * (edges[1] - edges[0])])
print("last 10 finishing tiles (entries, start, end):", sorted(zip(en.round(1), st.round(1), tc))[-10:])
for title, sel in (("last finishing", np.argsort(en)[-8:]), ("longest", np.argsort(du)[-8:])):
    print(title)
    for q in sel:
        print("   tile %5d entries %5d start %.1f dur %.1f: " % (q, tc[q], st[q], du[q]) + ", ".join("%s %.2f" % (n, (tt[q, b] - tt[q, a]) / 100.0) for n, (a, b) in zip(names, idx)))
A = np.vstack([np.ones_like(tc), tc]).T.astype(float)
coef = np.linalg.lstsq(A, du, rcond=None)[0]
print("fit: dur = %.2f us + %.4f us * entries" % tuple(coef))
A3 = np.hstack([np.ones((len(tp), 1)), tp])
c3 = np.linalg.lstsq(A3, du, rcond=None)[0]
print("fit: dur = %.2f us + %.4f * c0 + %.4f * c1 + %.4f * c2 + %.4f * c3 + %.4f * multi; residual rms %.2f us (entries-only fit: %.2f us)" % (
    *c3, np.sqrt(np.mean((A3 @ c3 - du) ** 2)), np.sqrt(np.mean((A @ coef - du) ** 2))))
