"""Helpers shared by the parity tests: run one scene through the oracle and through the HIP library."""
import numpy as np

from renderer_amd import api


def scene_matrices(orc, scene, width, height):
    models = orc.to_model(scene.transforms)
    view, proj = orc.camera(scene.camera, width, height)
    return models, view, proj


def render_oracle(orc, scene, width=None, height=None, models=None, view=None, proj=None, **kw):
    width = width or scene.size[0]
    height = height or scene.size[1]
    m, v, p = scene_matrices(orc, scene, width, height)
    return orc.render(scene, m if models is None else models, v if view is None else view,
                      p if proj is None else proj, width, height, **kw)


def render_hip(scene, width=None, height=None, models=None, view=None, proj=None, shading=0, scissor=None,
               want_clip=False, use_drawables=False, image=0, light=None, reference_indexing=False, bgra=False):
    width = width or scene.size[0]
    height = height or scene.size[1]
    with api.HipDrawer(width, height, coverage=True, shading=shading, scissor=scissor, reference_indexing=reference_indexing,
                       bgra=bgra) as d:
        if use_drawables:
            d.set_textures(scene.textures)
            d.update_drawables(scene.drawables())
        else:
            d.update_scene(scene)
        if light is not None:
            d.set_light(**light)
        if models is not None:
            d.update_uniforms(models)
        if view is not None:
            d.update_camera_matrices(view, proj)
        else:
            d.update_camera(scene.camera)
        d.draw(image)
        d.sync()
        out = {"color": d.read_color(image), "depth": d.read_depth(image), "prim": d.read_coverage(image),
               "stats": d.stats()}
        if want_clip:
            out["clip"] = d.read_clip_positions()
        return out


def assert_frames_equal(a, b, rows=None, what=""):
    """coverage and depth bit-exact, colour bit-exact (the stated bar is +-1 LSB; the design reaches 0)."""
    sl = slice(None) if rows is None else slice(rows[0], rows[1])
    pa, pb = a["prim"][sl], b["prim"][sl]
    nbad = int((pa != pb).sum())
    assert nbad == 0, "%s coverage differs at %d pixels, first %s" % (what, nbad, np.argwhere(pa != pb)[:4].tolist())
    da, db = a["depth"][sl].view(np.uint32), b["depth"][sl].view(np.uint32)
    nbad = int((da != db).sum())
    assert nbad == 0, "%s depth bits differ at %d pixels, first %s" % (what, nbad, np.argwhere(da != db)[:4].tolist())
    ca, cb = a["color"][sl].astype(np.int16), b["color"][sl].astype(np.int16)
    diff = np.abs(ca - cb).max() if ca.size else 0
    assert diff == 0, "%s colour differs by up to %d LSB at %d pixels" % (what, diff, int((ca != cb).any(axis=-1).sum()))


def soup_scene(positions, uvs=None, colors=None, textures=None, width=64, height=64, objects_split=None, tex_ids=None):
    """A scene whose vertices are already in clip space (w = 1): render it with identity
    matrices.  positions: (n_tri, 3, 3) float32 (x, y in NDC, z in [0,1] visible).
    objects_split: list of triangle counts per object (default one object)."""
    from renderer_amd import scenes
    from renderer_amd.api import MESH, OBJECT, VERTEX

    positions = np.asarray(positions, dtype=np.float32).reshape(-1, 3, 3)
    n = len(positions)
    v = np.zeros(3 * n, dtype=VERTEX)
    v["pos"] = positions.reshape(-1, 3)
    v["uv"] = np.tile(np.array([[0, 0], [1, 0], [0, 1]], dtype=np.float32), (n, 1)) if uvs is None else np.asarray(uvs, np.float32).reshape(-1, 2)
    v["color"] = 1.0 if colors is None else np.asarray(colors, np.float32).reshape(-1, 3)
    idx = np.arange(3 * n, dtype=np.uint32)
    split = [n] if objects_split is None else list(objects_split)
    assert sum(split) == n
    meshes, first = [], 0
    for k in split:
        meshes.append((3 * first, 3 * k, 3 * first, 3 * k))
        first += k
    meshes = np.array(meshes, dtype=MESH)
    # indices are mesh-local
    for (fv, nv, fi, ni) in meshes:
        idx[fi:fi + ni] -= fv
    objects = np.zeros(len(split), dtype=OBJECT)
    objects["mesh"] = np.arange(len(split))
    objects["texture"] = 0 if tex_ids is None else tex_ids
    objects["rendered"] = 1
    tr = api.identity_transforms(1).repeat(len(split))
    texs = [scenes.checker_texture(64, 5)] if textures is None else textures
    return scenes.Scene(v, idx, meshes, objects, np.ascontiguousarray(tr), texs, size=(width, height), name="soup")


def px_to_ndc(p, size):
    """Pixel coordinate (may be fractional) -> NDC so that the viewport map returns it exactly
    (size must be a power of two for exactness)."""
    return 2.0 * np.asarray(p, dtype=np.float64) / size - 1.0


IDENT = np.eye(4, dtype=np.float32).reshape(16)


def random_soup(rng, n, width, height, kinds=("tiny", "small", "medium", "large", "offscreen", "degenerate", "near", "huge")):
    """Seeded triangle soup in clip space (w = 1) with the awkward cases the rasteriser must survive."""
    tris = np.zeros((n, 3, 3), dtype=np.float32)
    for i in range(n):
        kind = kinds[rng.integers(len(kinds))]
        c = rng.uniform(-1.1, 1.1, 2)
        z = rng.uniform(0.05, 0.95, 3)
        if kind == "tiny":
            r = 2.0 / width
        elif kind == "small":
            r = 6.0 / width
        elif kind == "medium":
            r = 0.2
        elif kind == "large":
            r = 1.5
        elif kind == "offscreen":
            c = rng.uniform(1.5, 3.0, 2) * rng.choice([-1, 1], 2)
            r = 0.4
        elif kind == "degenerate":
            r = 0.3
        elif kind == "near":
            r = 0.6
            z = rng.uniform(-0.5, 1.5, 3)  # crosses z = 0 and / or z = w
        else:  # huge: forces the guard-band clipper
            r = rng.uniform(5.0, 40.0)
        p = c[None, :] + rng.uniform(-r, r, (3, 2))
        if kind == "degenerate":
            p[2] = p[0] if rng.random() < 0.5 else 0.5 * (p[0] + p[1])
        tris[i, :, :2] = p
        tris[i, :, 2] = z
    # snap some vertices onto exact pixel centres / corners to exercise the fill rule
    m = rng.random((n, 3)) < 0.3
    snap = np.round((tris[..., :2] + 1) * 0.5 * np.array([width, height]) * 2) / 2
    tris[..., :2][m] = (snap[m] / np.array([width, height]) * 2 - 1).astype(np.float32)
    return tris


def write_scene_file(path, scene, n_images=2, n_frames=1, rotate_deg=(0.0, 0.0, 0.0), texture_files=None):
    """Scene file of host/engine_demo.cpp (format described there): reference-style drawables, one
    mesh copy per object, textures referred to by path.  With `texture_files` (one file name per
    texture, relative to the scene file) the texels are left out and the program loads the files."""
    import struct
    w, h = scene.size
    with open(path, "wb") as f:
        f.write(b"SXES" + struct.pack("<I", 1))
        f.write(struct.pack("<4I", w, h, n_images, n_frames))
        f.write(np.asarray(rotate_deg, dtype="<f4").tobytes())
        f.write(np.ascontiguousarray(scene.camera).tobytes())
        f.write(struct.pack("<I", len(scene.textures)))
        for k, t in enumerate(scene.textures):
            name = (texture_files[k] if texture_files else "textures/tex%03d.jpg" % k).encode()
            f.write(struct.pack("<I", len(name)) + name)
            if texture_files:
                f.write(struct.pack("<2I", 0, 0))
                continue
            f.write(struct.pack("<2I", t.shape[1], t.shape[0]))
            f.write(np.ascontiguousarray(t, dtype=np.uint8).tobytes())
        f.write(struct.pack("<I", len(scene.objects)))
        for o, t in zip(scene.objects, scene.transforms):
            m = scene.meshes[o["mesh"]]
            v = scene.vertices[m["first_vertex"]:m["first_vertex"] + m["n_vertices"]]
            i = scene.indices[m["first_index"]:m["first_index"] + m["n_indices"]]
            f.write(struct.pack("<2I", len(v), len(i)))
            f.write(np.ascontiguousarray(v).tobytes())
            f.write(np.ascontiguousarray(i, dtype="<u4").tobytes())
            f.write(np.ascontiguousarray(t).tobytes())
            f.write(struct.pack("<iI", int(o["texture"]), int(o["rendered"])))
