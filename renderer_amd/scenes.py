"""Deterministic synthetic scenes for the raster/shade path (inputs only: no rendering here).

The reference ships one concrete mesh, `Cube::new` (src/primitives/three_d/cube.rs:12-83), builds
its world from 64 randomly placed cubes (src/main.rs:201-227) and cannot hold more than 1024
objects (src/renderer/allocator.rs:311-318); its textures are git-LFS pointers.  The benchmark
configurations therefore use the cube, a 1,024-triangle grid sphere, and procedural opaque RGBA8
textures, all generated from fixed seeds.
"""
import numpy as np

from . import api
from .api import MESH, OBJECT, TRANSFORM, VERTEX


class Scene:
    def __init__(self, vertices, indices, meshes, objects, transforms, textures, camera=None, size=(800, 600),
                 name="scene"):
        self.vertices = np.ascontiguousarray(vertices, dtype=VERTEX)
        self.indices = np.ascontiguousarray(indices, dtype=np.uint32)
        self.meshes = np.ascontiguousarray(meshes, dtype=MESH)
        self.objects = np.ascontiguousarray(objects, dtype=OBJECT)
        self.transforms = np.ascontiguousarray(transforms, dtype=TRANSFORM)
        self.textures = [np.ascontiguousarray(t, dtype=np.uint8) for t in textures]
        self.camera = api.identity_transforms(1) if camera is None else np.ascontiguousarray(camera, dtype=TRANSFORM).reshape(1)
        self.size = size
        self.name = name

    @property
    def n_triangles(self):
        o = self.objects
        return int((self.meshes["n_indices"][o["mesh"]][o["rendered"] != 0] // 3).sum())

    def drawables(self):
        """The same scene as a list of reference-style drawables (one mesh copy per object)."""
        out = []
        for o, t in zip(self.objects, self.transforms):
            m = self.meshes[o["mesh"]]
            v = self.vertices[m["first_vertex"]:m["first_vertex"] + m["n_vertices"]]
            i = self.indices[m["first_index"]:m["first_index"] + m["n_indices"]]
            out.append(api.Drawable(v, i, t, int(o["texture"]), bool(o["rendered"])))
        return out


def cube_mesh():
    """Cube::new: 24 vertices, 36 indices, +-0.5, per-face colour and uv (cube.rs:28-73)."""
    lo, hi = -0.5, 0.5
    red, green, blue = (1, 0, 0), (0, 1, 0), (0, 0, 1)
    yellow, purple, blue_green = (1, 1, 0), (1, 0, 1), (0, 1, 1)
    uv = [(0, 0), (1, 0), (1, 1), (0, 1)]
    faces = [
        (red, [(lo, lo, lo), (lo, hi, lo), (hi, hi, lo), (hi, lo, lo)]),          # back
        (green, [(lo, lo, hi), (hi, lo, hi), (hi, hi, hi), (lo, hi, hi)]),        # front
        (blue, [(lo, lo, hi), (lo, hi, hi), (lo, hi, lo), (lo, lo, lo)]),         # left
        (yellow, [(hi, lo, hi), (hi, lo, lo), (hi, hi, lo), (hi, hi, hi)]),       # right
        (purple, [(lo, hi, hi), (hi, hi, hi), (hi, hi, lo), (lo, hi, lo)]),       # top
        (blue_green, [(lo, lo, hi), (lo, lo, lo), (hi, lo, lo), (hi, lo, hi)]),   # bottom
    ]
    v = np.zeros(24, dtype=VERTEX)
    k = 0
    for colour, corners in faces:
        for c, t in zip(corners, uv):
            v[k]["pos"], v[k]["color"], v[k]["uv"] = c, colour, t
            k += 1
    idx = []
    for f in range(6):
        b = 4 * f
        idx += [b, b + 1, b + 2, b + 2, b + 3, b]
    return v, np.array(idx, dtype=np.uint32)


def grid_sphere_mesh(n_lon=32, n_lat=16, radius=0.5):
    """(n_lon+1) x (n_lat+1) grid wrapped on a sphere: 561 vertices / 1,024 triangles by default.
    Pole cells contain zero-area triangles on purpose (the setup stage must cull them)."""
    i = np.arange(n_lon + 1)
    j = np.arange(n_lat + 1)
    ii, jj = np.meshgrid(i, j, indexing="xy")
    theta = 2.0 * np.pi * ii / n_lon
    phi = np.pi * jj / n_lat
    v = np.zeros((n_lat + 1) * (n_lon + 1), dtype=VERTEX)
    v["pos"][:, 0] = (radius * np.sin(phi) * np.cos(theta)).ravel()
    v["pos"][:, 1] = (radius * np.cos(phi)).ravel()
    v["pos"][:, 2] = (radius * np.sin(phi) * np.sin(theta)).ravel()
    v["uv"][:, 0] = (ii / n_lon).ravel()
    v["uv"][:, 1] = (jj / n_lat).ravel()
    h = (ii * 73856093 ^ jj * 19349663).ravel().astype(np.uint64)
    v["color"][:, 0] = ((h >> 0) & 255) / 255.0
    v["color"][:, 1] = ((h >> 8) & 255) / 255.0
    v["color"][:, 2] = ((h >> 16) & 255) / 255.0
    idx = []
    w = n_lon + 1
    for y in range(n_lat):
        for x in range(n_lon):
            a, b, c, d = y * w + x, y * w + x + 1, (y + 1) * w + x + 1, (y + 1) * w + x
            idx += [a, b, c, c, d, a]
    return v, np.array(idx, dtype=np.uint32)


def checker_texture(size, seed):
    """Opaque RGBA8 (sRGB-encoded) checker of 8x8-texel blocks in two seeded colours with a
    gradient across u and v."""
    rng = np.random.default_rng(seed)
    ca = rng.integers(64, 256, size=3)
    cb = rng.integers(0, 160, size=3)
    y, x = np.mgrid[0:size, 0:size]
    chk = ((x // 8 + y // 8) & 1).astype(bool)
    t = np.zeros((size, size, 4), dtype=np.uint8)
    gx = 96 + (x * 159) // max(size - 1, 1)
    gy = 96 + (y * 159) // max(size - 1, 1)
    for k in range(3):
        g = gx if k != 1 else gy
        t[..., k] = np.where(chk, (ca[k] * g) // 255, (cb[k] * g) // 255)
    t[..., 3] = 255
    return t


def translucent_texture(size, seed):
    """The checker of checker_texture with alpha: one colour of the checker half transparent, a
    fully transparent stripe, and alpha ramps in between (every blending regime in one texture)."""
    t = checker_texture(size, seed)
    y, x = np.mgrid[0:size, 0:size]
    chk = ((x // 8 + y // 8) & 1).astype(bool)
    alpha = np.where(chk, 128, 255)
    alpha = np.where((y * 4 // size) == 1, (x * 255) // max(size - 1, 1), alpha)  # a ramp from clear to solid
    alpha = np.where((x * 8 // size) == 5, 0, alpha)                               # a clear stripe
    t[..., 3] = alpha.astype(np.uint8)
    return t


def _single_mesh_scene(v, idx, transforms, textures_of, textures, **kw):
    meshes = np.array([(0, len(v), 0, len(idx))], dtype=MESH)
    objects = np.zeros(len(transforms), dtype=OBJECT)
    objects["mesh"] = 0
    objects["texture"] = textures_of
    objects["rendered"] = 1
    return Scene(v, idx, meshes, objects, transforms, textures, **kw)


def cube_scene(width=800, height=600):
    """Config 1: one textured cube at (0,0,-3) rotated by rotate(30,45,0); identity camera."""
    v, idx = cube_mesh()
    t = api.identity_transforms(1)
    api.translate(t, 0, 0.0, 0.0, -3.0)
    api.rotate(t, 0, 30.0, 45.0, 0.0)
    return _single_mesh_scene(v, idx, t, 0, [checker_texture(256, 1)], size=(width, height), name="cfg1-cube")


def reference_demo_scene(n=64, seed=7, width=1024, height=768):
    """The world `start_engine` builds (src/main.rs:152-163,201-227): n cubes uniformly in
    [-15,15]^3 with one of three textures, seen by an identity camera; seeded instead of thread_rng."""
    v, idx = cube_mesh()
    rng = np.random.default_rng(seed)
    t = api.identity_transforms(n)
    pos = rng.uniform(-15.0, 15.0, size=(n, 3)).astype(np.float32)
    for i in range(n):
        api.translate(t, i, float(pos[i, 0]), float(pos[i, 1]), float(pos[i, 2]))
    tex = rng.integers(0, 3, size=n)
    return _single_mesh_scene(v, idx, t, tex, [checker_texture(256, 10 + k) for k in range(3)],
                              size=(width, height), name="reference-demo")


def sphere_field_scene(n, seed, box_xy, z_range, width, height, n_textures=4, tex_size=512, name="spheres"):
    """n instances of the 1,024-triangle grid sphere, uniformly placed in
    [-box_xy,box_xy]^2 x [z_range], uniform scale in [0.5,1.5], random Euler rotation."""
    v, idx = grid_sphere_mesh()
    rng = np.random.default_rng(seed)
    t = api.identity_transforms(1).repeat(n)
    pos = np.empty((n, 3), dtype=np.float32)
    pos[:, 0:2] = rng.uniform(-box_xy, box_xy, size=(n, 2))
    pos[:, 2] = rng.uniform(z_range[0], z_range[1], size=n)
    scale = rng.uniform(0.5, 1.5, size=n).astype(np.float32)
    euler = rng.uniform(0.0, 360.0, size=(n, 3)).astype(np.float32)
    t["position"] = pos
    t["scale"] = scale[:, None]
    t = np.ascontiguousarray(t)
    for i in range(n):
        api.rotate(t, i, float(euler[i, 0]), float(euler[i, 1]), float(euler[i, 2]))
    tex = rng.integers(0, n_textures, size=n)
    return _single_mesh_scene(v, idx, t, tex, [checker_texture(tex_size, 100 + k) for k in range(n_textures)],
                              size=(width, height), name=name)


def mixed_scene(n=40, seed=21, width=640, height=360):
    """Two different meshes (the cube and the grid sphere) spread over three texture groups, some
    objects with `rendered` false (src/components/mesh.rs:8): the draw order the reference derives
    (texture groups ascending, drawable order inside a group, src/renderer/drawer.rs:313-347,481-487)
    then differs from the caller's order, and the meshes have different index counts."""
    cv, ci = cube_mesh()
    sv, si = grid_sphere_mesh()
    vertices = np.concatenate([cv, sv])
    indices = np.concatenate([ci, si])
    meshes = np.array([(0, len(cv), 0, len(ci)), (len(cv), len(sv), len(ci), len(si))], dtype=MESH)
    rng = np.random.default_rng(seed)
    t = np.ascontiguousarray(api.identity_transforms(1).repeat(n))
    t["position"][:, 0:2] = rng.uniform(-6.0, 6.0, size=(n, 2))
    t["position"][:, 2] = rng.uniform(-18.0, -4.0, size=n)
    t["scale"] = rng.uniform(0.6, 2.0, size=(n, 1)).astype(np.float32)
    euler = rng.uniform(0.0, 360.0, size=(n, 3)).astype(np.float32)
    for i in range(n):
        api.rotate(t, i, float(euler[i, 0]), float(euler[i, 1]), float(euler[i, 2]))
    objects = np.zeros(n, dtype=OBJECT)
    objects["mesh"] = rng.integers(0, 2, size=n)
    objects["texture"] = rng.integers(0, 3, size=n)
    objects["rendered"] = (rng.random(n) > 0.25).astype(np.uint32)
    return Scene(vertices, indices, meshes, objects, t, [checker_texture(128, 40 + k) for k in range(3)],
                 size=(width, height), name="mixed-meshes-groups")


class Quad:
    """The 2-D UI primitive (src/primitives/two_d/quad.rs:12-60): a rectangle given by how far its
    sides lie from the centre of its parent, in units where the parent spans -1..1 (no parent: the
    whole target in NDC).  Drawn by the UI pipeline: positions passed through unchanged
    (ui.vert:11-15), vertex colour times texture (ui.frag:12)."""

    def __init__(self, key, top, left, bottom, right, parent=None, rendered=True):
        self.key, self.rendered, self.children = key, rendered, []
        # calculate_absolute_vertices (quad.rs:76-121); float32 step by step, as the reference computes
        f = np.float32
        if parent is None:
            plx, pux, ply, puy = f(-1.0), f(1.0), f(-1.0), f(1.0)
        else:
            lo, up = parent.vertices[0]["pos"], parent.vertices[2]["pos"]
            plx, pux, ply, puy = f(lo[0]), f(up[0]), f(lo[1]), f(up[1])
        x_range = f(pux - plx)
        x_half = f(f(x_range / f(2.0)) + plx)
        y_range = f(puy - ply)
        y_half = f(f(y_range / f(2.0)) + ply)

        def dist(rng, rel):  # calculate_normalized_distance: |rel / 2 * range|
            return f(abs(f(f(f(rel) / f(2.0)) * rng)))
        lower_x, upper_x = f(x_half - dist(x_range, left)), f(x_half + dist(x_range, right))
        lower_y, upper_y = f(y_half - dist(y_range, bottom)), f(y_half + dist(y_range, top))
        self.vertices = np.zeros(4, dtype=VERTEX)
        corners = [(lower_x, lower_y), (upper_x, lower_y), (upper_x, upper_y), (lower_x, upper_y)]
        colours = [(1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 1)]
        uvs = [(0, 0), (1, 0), (1, 1), (0, 1)]
        for k in range(4):
            self.vertices[k]["pos"] = (corners[k][0], corners[k][1], 0.0)
            self.vertices[k]["color"] = colours[k]
            self.vertices[k]["uv"] = uvs[k]
        self.indices = np.array([0, 1, 2, 2, 3, 0], dtype=np.uint32)

    def with_children(self, children):
        self.children = list(children)
        return self

    def quads(self):
        """This quad and its descendants in the order Quad::meshes lists them (quad.rs:123-136)."""
        out = [self]
        for c in self.children:
            out += c.quads()
        return out


def esc_menu():
    """EscMenu::new (src/primitives/two_d/widget.rs:16-26): an unrendered box with two children."""
    main_box = Quad("escmenu", 0.5, -0.5, 0.5, 0.5)
    child1 = Quad("escmenu_child1", 0.7, -0.85, 0.3, 0.85, main_box)
    child2 = Quad("escmenu_child2", 0.0, -0.5, -0.9, 0.3, main_box)
    main_box.rendered = False
    return main_box.with_children([child1, child2])


def ui_scene(root, textures, texture_of=None, width=320, height=240):
    """A widget's quads as drawables for the UI pipeline: one mesh per quad (Quad::meshes), identity
    transforms; render it with identity view and projection matrices and the colour-times-texture
    shading mode, which is what ui.vert / ui.frag compute."""
    quads = root.quads()
    vertices = np.concatenate([q.vertices for q in quads])
    indices = np.concatenate([q.indices for q in quads])  # mesh-local
    meshes = np.array([(4 * i, 4, 6 * i, 6) for i in range(len(quads))], dtype=MESH)
    objects = np.zeros(len(quads), dtype=OBJECT)
    objects["mesh"] = np.arange(len(quads))
    objects["texture"] = 0 if texture_of is None else texture_of
    objects["rendered"] = [1 if q.rendered else 0 for q in quads]
    t = np.ascontiguousarray(api.identity_transforms(1).repeat(len(quads)))
    return Scene(vertices, indices, meshes, objects, t, textures, size=(width, height), name="ui-quads")


def config_scene(cfg, scale=1.0):
    """The benchmark configurations. `scale` < 1 shrinks object count and frame size together
    for parity tests the CPU oracle can finish in seconds."""
    if cfg == 1:
        return cube_scene(max(8, int(800 * scale)), max(8, int(600 * scale)))
    if cfg == 2:
        return sphere_field_scene(max(1, int(100 * scale)), 2, 15.0, (-40.0, -5.0), max(8, int(1920 * scale)),
                                  max(8, int(1080 * scale)), name="cfg2-100-spheres-1080p")
    if cfg in (3, 4):
        w, h = (3840, 2160) if cfg == 3 else (7680, 4320)
        return sphere_field_scene(max(1, int(10000 * scale)), 3, 60.0, (-200.0, -5.0), max(8, int(w * scale)),
                                  max(8, int(h * scale)), name="cfg%d-10k-spheres" % cfg)
    if cfg == 5:
        s = sphere_field_scene(max(1, int(100 * scale)), 2, 15.0, (-40.0, -5.0), max(8, int(512 * scale)),
                               max(8, int(512 * scale)), name="cfg5-multiview")
        return s
    raise ValueError("unknown config %r" % (cfg,))


def orbit_cameras(n, seed, radius=60.0, target=(0.0, 0.0, -22.5)):
    """n camera Transforms on a seeded sphere looking at `target`, expressed so that
    update_camera (euler.y -> pitch, euler.x -> yaw, src/renderer/drawer.rs:244-246) reproduces them."""
    rng = np.random.default_rng(seed)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[:, 1] = np.clip(d[:, 1], -0.95, 0.95)  # keep |pitch| away from the gimbal branch
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    cams = api.identity_transforms(1).repeat(n)
    cams = np.ascontiguousarray(cams)
    tgt = np.asarray(target, dtype=np.float64)
    for i in range(n):
        eye = tgt + radius * d[i]
        b = d[i]  # camera backward axis
        pitch = -np.degrees(np.arcsin(b[1]))
        yaw = np.degrees(np.arctan2(b[0], b[2]))
        cams[i]["position"] = eye.astype(np.float32)
        api.rotate(cams, i, float(yaw), float(pitch), 0.0)
    return cams
