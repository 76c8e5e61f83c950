#!/usr/bin/env python3
"""Development helper (CPU only): distribution of per-tile pixel-box sizes of config 3's triangles."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from renderer_amd import api, scenes
s = scenes.config_scene(3); w, h = s.size
m = api.to_model(s.transforms).reshape(-1, 4, 4).astype(np.float64)  # the library's host-side maths
v, p = api.camera_matrices(s.camera, w, h)
V = np.array(v, dtype=np.float64).reshape(4, 4).T; P = np.array(p, dtype=np.float64).reshape(4, 4).T
PVM = P @ V @ np.transpose(m, (0, 2, 1))
nv = int(s.meshes[0]['n_vertices']); nt = int(s.meshes[0]['n_triangles']) if 'n_triangles' in s.meshes.dtype.names else 1024
pos = np.concatenate([s.vertices['pos'][:nv].astype(np.float64), np.ones((nv, 1))], 1)
idx = s.indices[:3 * nt].reshape(-1, 3)
sizes = []
for o in range(len(PVM)):
    c = pos @ PVM[o].T
    wv = c[:, 3]
    ok = (wv > 1e-6)
    x = (c[:, 0] / wv + 1) * w / 2; y = (c[:, 1] / wv + 1) * h / 2
    tx = x[idx]; ty = y[idx]; tok = ok[idx].all(1)
    inside = tok & (tx.max(1) >= 0) & (tx.min(1) < w) & (ty.max(1) >= 0) & (ty.min(1) < h)
    bx0 = np.ceil(tx.min(1) - 0.5); bx1 = np.floor(tx.max(1) - 0.5); by0 = np.ceil(ty.min(1) - 0.5); by1 = np.floor(ty.max(1) - 0.5)
    bx0 = np.maximum(bx0, 0); bx1 = np.minimum(bx1, w - 1); by0 = np.maximum(by0, 0); by1 = np.minimum(by1, h - 1)
    good = inside & (bx1 >= bx0) & (by1 >= by0)
    sizes.append(((bx1 - bx0 + 1) * (by1 - by0 + 1))[good])
sz = np.concatenate(sizes)
print("triangles with non-empty box", len(sz), "total box px", sz.sum())
edges = [1, 2, 3, 5, 9, 17, 33, 65, 129, 1025, 10**9]
for a, b in zip(edges[:-1], edges[1:]):
    msk = (sz >= a) & (sz < b)
    print("box px [%d,%d): %8d tris (%.1f%%), %9d px (%.1f%%)" % (a, b, msk.sum(), 100 * msk.mean(), sz[msk].sum(), 100 * sz[msk].sum() / sz.sum()))
