#!/usr/bin/env python3
"""Regenerates the golden fixtures from the CPU oracle:  python tests/golden/make_golden.py [fixture.npz ...]

The reference ships no golden images and cannot run offline, so these vectors are outputs of the
oracle (itself pinned by tests/test_oracle_math.py and tests/test_oracle_raster.py); they freeze
its behaviour so that both the oracle and the HIP library are checked against fixed bits.
Each fixture stores its inputs as well, so no random generator is involved when it is replayed.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import oracle  # noqa: E402
from renderer_amd import scenes  # noqa: E402
from tests.util import IDENT, random_soup, soup_scene  # noqa: E402


def save(name, scene, models, view, proj, out, extra=None):
    d = dict(vertices=scene.vertices, indices=scene.indices, meshes=scene.meshes, objects=scene.objects,
             transforms=scene.transforms, camera=scene.camera, size=np.array(scene.size),
             models=models, view=view, proj=proj, color=out["color"], depth_bits=out["depth"].view(np.uint32),
             prim=out["prim"])
    for i, t in enumerate(scene.textures):
        d["texture%d" % i] = t
    d.update(extra or {})
    if len(sys.argv) > 1 and name not in sys.argv[1:]:
        return  # (only the fixtures named on the command line are rewritten)
    np.savez_compressed(os.path.join(HERE, name), **d)
    print(name, out["stats"])


def main():
    oracle.build()
    # 1. the reference's cube (config 1 pose) at 64x48
    s = scenes.cube_scene(64, 48)
    m = oracle.to_model(s.transforms)
    v, p = oracle.camera(s.camera, 64, 48)
    save("cube_64x48.npz", s, m, v, p, oracle.render(s, m, v, p, 64, 48))
    # 2. the 64-cube demo world at 160x120
    s = scenes.reference_demo_scene(width=160, height=120)
    m = oracle.to_model(s.transforms)
    v, p = oracle.camera(s.camera, 160, 120)
    save("demo_160x120.npz", s, m, v, p, oracle.render(s, m, v, p, 160, 120))
    # 3. a clip-space triangle soup with every awkward case, identity matrices
    rng = np.random.default_rng(21)
    s = soup_scene(random_soup(rng, 500, 96, 96), width=96, height=96)
    m = np.tile(IDENT, (1, 1))
    save("soup_96x96.npz", s, m, IDENT, IDENT, oracle.render(s, m, IDENT, IDENT, 96, 96))
    # 4. spheres seen by a rotated, translated camera (exercises fps_view and Euler extraction)
    s = scenes.sphere_field_scene(12, 8, 6.0, (-20.0, -4.0), 128, 96, n_textures=2, tex_size=32)
    cam = oracle.identity_transforms(1)
    oracle.translate(cam, 0, 1.5, -0.5, 2.0)
    oracle.rotate(cam, 0, 12.0, -7.0, 0.0)
    s.camera = cam
    m = oracle.to_model(s.transforms)
    v, p = oracle.camera(cam, 128, 96)
    save("spheres_cam_128x96.npz", s, m, v, p, oracle.render(s, m, v, p, 128, 96))
    # 5. the cube with a translucent texture: fragments blended in submission order (BlendState::ALPHA)
    s = scenes.cube_scene(96, 72)
    s.textures = [scenes.translucent_texture(64, 3)]
    m = oracle.to_model(s.transforms)
    v, p = oracle.camera(s.camera, 96, 72)
    save("blend_cube_96x72.npz", s, m, v, p, oracle.render(s, m, v, p, 96, 72))
    # 6. one triangle through the near and the far plane whose fan triangles, snapped corner by corner,
    # overlap by a sliver at pixel (33, 51): the later fan triangle has the smaller depth there and stays
    tri = np.array([[[0.55813956, -0.48756218, -0.02677643], [0.5810299, -0.33630487, 0.6650742],
                     [0.49605954, -0.896145, 1.0457937]]], dtype=np.float32)
    uv = np.array([[0.04350607, -0.09129993], [0.7519953, 1.4499753], [-0.29209238, 0.08491877]], dtype=np.float32)
    s = soup_scene(tri, uvs=uv, textures=[scenes.checker_texture(16, 3)], width=43, height=201)
    m = np.tile(IDENT, (1, 1))
    save("sliver_soup_43x201.npz", s, m, IDENT, IDENT, oracle.render(s, m, IDENT, IDENT, 43, 201))
    # 7. the 64-cube / 3-texture demo world (src/main.rs:201-227) as the reference binary records its
    # draws (src/renderer/drawer.rs:313-347): the matrix index restarts with every texture group
    s = scenes.reference_demo_scene(seed=11, width=256, height=192)  # (a seed under which the recorded draws show cubes)
    m = oracle.to_model(s.transforms)
    v, p = oracle.camera(s.camera, 256, 192)
    save("demo_reference_indexing_256x192.npz", s, m, v, p,
         oracle.render(s, m, v, p, 256, 192, flags=oracle.FLAG_REFERENCE_INDEXING), extra={"reference_indexing": np.array(1)})


if __name__ == "__main__":
    main()
