#!/usr/bin/env python3
"""Development helper (CPU only): how many lane-walk iterations the tile kernel's waves make on
config 3 for different ways of ordering a tile's bin by pixel-box size.

A wave takes 64 consecutive bin entries; its lanes walk one pixel box each, so the wave makes as
many iterations as its largest box has pixels.  Boxes above 24 pixels go to the shared queue and
are left out.  Orders compared: size classes with the given upper bounds (random inside a class),
and a full sort by box size.
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from renderer_amd import api, scenes

s = scenes.config_scene(3); w, h = s.size
m = api.to_model(s.transforms).reshape(-1, 4, 4).astype(np.float64)
v, p = api.camera_matrices(s.camera, w, h)
V = np.array(v, dtype=np.float64).reshape(4, 4).T; P = np.array(p, dtype=np.float64).reshape(4, 4).T
PVM = P @ V @ np.transpose(m, (0, 2, 1))
nv = int(s.meshes[0]['n_vertices']); nt = int(s.meshes[0]['n_indices']) // 3
pos = np.concatenate([s.vertices['pos'][:nv].astype(np.float64), np.ones((nv, 1))], 1)
idx = s.indices[:3 * nt].reshape(-1, 3)
tiles, sizes = [], []
for o in range(len(PVM)):
    c = pos @ PVM[o].T
    wv = c[:, 3]
    ok = wv > 1e-6
    x = (c[:, 0] / np.where(ok, wv, 1) + 1) * w / 2; y = (c[:, 1] / np.where(ok, wv, 1) + 1) * h / 2
    tx = x[idx]; ty = y[idx]; tok = ok[idx].all(1)
    area = (tx[:, 1] - tx[:, 0]) * (ty[:, 2] - ty[:, 0]) - (tx[:, 2] - tx[:, 0]) * (ty[:, 1] - ty[:, 0])
    bx0 = np.maximum(np.ceil(tx.min(1) - 0.5), 0); bx1 = np.minimum(np.floor(tx.max(1) - 0.5), w - 1)
    by0 = np.maximum(np.ceil(ty.min(1) - 0.5), 0); by1 = np.minimum(np.floor(ty.max(1) - 0.5), h - 1)
    good = tok & (bx1 >= bx0) & (by1 >= by0) & (np.abs(area) > 1e-9)
    one = good & (bx0 // 32 == bx1 // 32) & (by0 // 32 == by1 // 32)  # single-tile triangles only (95 % of the entries)
    tiles.append(((by0 // 32) * ((w + 31) // 32) + bx0 // 32)[one].astype(np.int64))
    sizes.append(((bx1 - bx0 + 1) * (by1 - by0 + 1))[one].astype(np.int64))
tile = np.concatenate(tiles); sz = np.concatenate(sizes)
keep = sz <= 24
tile, sz = tile[keep], sz[keep]
print("entries", len(sz), "box pixels", sz.sum(), "-> ideal wave iterations", sz.sum() / 64)
rng = np.random.default_rng(0)


def iterations(key):
    """entries ordered by (tile, key, random); waves of 64 consecutive entries inside a tile"""
    order = np.lexsort((rng.random(len(sz)), key, tile))
    t, b = tile[order], sz[order]
    first = np.r_[0, np.flatnonzero(np.diff(t)) + 1]
    within = np.arange(len(t)) - np.repeat(first, np.diff(np.r_[first, len(t)]))
    wave = np.cumsum(np.r_[1, (np.diff(t) != 0) | (within[1:] % 64 == 0)])
    return np.maximum.reduceat(b, np.flatnonzero(np.r_[1, np.diff(wave)])).sum()


for bounds in ([2, 8, 24], [2, 4, 8, 24], [2, 4, 8, 14, 24], [2, 4, 6, 9, 14, 24], [1, 2, 4, 8, 12, 24], [2, 4, 6, 8, 12, 16, 24], [1, 2, 3, 4, 6, 8, 12, 16, 24]):
    cls = np.searchsorted(np.array(bounds), sz)
    print("classes <= %s: %d wave iterations" % (bounds, iterations(cls)))
print("sorted by box size: %d wave iterations" % iterations(sz))
print("unsorted: %d wave iterations" % iterations(np.zeros_like(sz)))
