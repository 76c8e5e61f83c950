"""The draw list as the reference binary records it (SXE_FLAG_COMPAT_REFERENCE_INDEXING /
ORC_FLAG_REFERENCE_INDEXING), src/renderer/drawer.rs:313-347:

  * inside a texture group the i-th mesh, unrendered ones counted, is drawn with the i-th model
    matrix of the whole scene (:326,331);
  * its index range starts where the ranges of the draws recorded before it end (:313,345-346),
    in an index buffer laid out in drawable order (:181-193);
  * an unrendered mesh does not advance that cursor (:327-329).

The tests restate those lines once more in numpy (`reference_draws`, `explicit_scene`): every
recorded draw becomes an object of its own, with the triangles its index range selects and the
transform its dynamic offset selects, and that scene is rendered the ordinary way.  The flagged
render of the original scene must give the same frame bit for bit."""
import numpy as np
import pytest

from renderer_amd import api, scenes
from renderer_amd.api import MESH, OBJECT
from tests.util import assert_frames_equal


def reference_draws(scene):
    """[(matrix index, texture, first index, index count)] in recording order."""
    groups = {}
    for o, obj in enumerate(scene.objects):  # BTreeMap<texture, Vec<&Mesh>>, drawer.rs:481-487
        groups.setdefault(int(obj["texture"]), []).append(o)
    cursor, draws = 0, []
    for tex in sorted(groups):
        for i, o in enumerate(groups[tex]):
            if not scene.objects[o]["rendered"]:
                continue
            n = int(scene.meshes[scene.objects[o]["mesh"]]["n_indices"])
            draws.append((i, tex, cursor, n))
            cursor += n
    return draws


def explicit_scene(scene, draws):
    """One object per recorded draw: the triangles of its index range, the transform of its matrix index."""
    cat_v, cat_i, base = [], [], 0
    for obj in scene.objects:  # generate_vertex_and_index_buffers, drawer.rs:173-193
        m = scene.meshes[obj["mesh"]]
        v = scene.vertices[m["first_vertex"]:m["first_vertex"] + m["n_vertices"]]
        i = scene.indices[m["first_index"]:m["first_index"] + m["n_indices"]]
        cat_v.append(v)
        cat_i.append(i.astype(np.int64) + base)
        base += len(v)
    cat_v, cat_i = np.concatenate(cat_v), np.concatenate(cat_i)
    vertices, indices, meshes = [], [], []
    objects = np.zeros(len(draws), dtype=OBJECT)
    transforms = np.zeros(len(draws), dtype=scene.transforms.dtype)
    nv = ni = 0
    for k, (matrix, tex, first, n) in enumerate(draws):
        idx = cat_i[first:first + n]
        lo, hi = (int(idx.min()), int(idx.max())) if n else (0, 0)
        v = cat_v[lo:hi + 1]
        vertices.append(v)
        indices.append((idx - lo).astype(np.uint32))
        meshes.append((nv, len(v), ni, n))
        nv += len(v)
        ni += n
        objects[k] = (k, tex, 1)
        transforms[k] = scene.transforms[matrix]
    return scenes.Scene(np.concatenate(vertices), np.concatenate(indices), np.array(meshes, dtype=MESH), objects,
                        transforms, scene.textures, camera=scene.camera, size=scene.size, name=scene.name + "-explicit")


def seven_cubes():
    """Seven cubes of different sizes (so that one can tell whose triangles were drawn) in a row,
    over three textures, the sixth unrendered."""
    cv, ci = scenes.cube_mesh()
    n = 7
    vertices, meshes = [], []
    for k in range(n):
        v = cv.copy()
        v["pos"] *= np.float32(0.5 + 0.15 * k)
        vertices.append(v)
        meshes.append((24 * k, 24, 36 * k, 36))
    t = np.ascontiguousarray(api.identity_transforms(1).repeat(n))
    for k in range(n):
        api.translate(t, k, -4.5 + 1.5 * k, 0.4 * ((k % 3) - 1), -9.0 - 0.25 * k)
        api.rotate(t, k, 20.0 + 7.0 * k, 30.0 + 11.0 * k, 0.0)
    objects = np.zeros(n, dtype=OBJECT)
    objects["mesh"] = np.arange(n)
    objects["texture"] = [2, 0, 1, 0, 2, 1, 1]
    objects["rendered"] = [1, 1, 1, 1, 1, 0, 1]
    return scenes.Scene(np.concatenate(vertices), np.tile(ci, n), np.array(meshes, dtype=MESH), objects, t,
                        [scenes.checker_texture(32, 60 + k) for k in range(3)], size=(320, 120), name="seven-cubes")


def hetero_scene():
    """Meshes of 36, 6 and 3072 indices over three texture groups with unrendered objects: index
    ranges that start inside another drawable's mesh and run over several."""
    s = scenes.mixed_scene(n=24, seed=5, width=320, height=200)
    # a third mesh: one quad (6 indices)
    q = scenes.Quad("q", 0.8, -0.8, 0.8, 0.8)
    qv = q.vertices.copy()
    qv["pos"][:, 2] = 0.1
    nv, ni = len(s.vertices), len(s.indices)
    s.vertices = np.concatenate([s.vertices, qv])
    s.indices = np.concatenate([s.indices, q.indices])
    s.meshes = np.concatenate([s.meshes, np.array([(nv, 4, ni, 6)], dtype=MESH)])
    s.objects["mesh"] = np.random.default_rng(17).integers(0, 3, size=len(s.objects))
    return s


def render_orc(orc, scene, compat, **kw):
    w, h = scene.size
    view, proj = orc.camera(scene.camera, w, h)
    return orc.render(scene, orc.to_model(scene.transforms), view, proj, w, h,
                      flags=orc.FLAG_REFERENCE_INDEXING if compat else 0, **kw)


def test_recorded_draws_of_seven_cubes_by_hand():
    # groups: texture 0 -> objects [1, 3], texture 1 -> [2, 5 (unrendered), 6], texture 2 -> [0, 4];
    # every mesh has 36 indices, so draw d reads the indices of drawable d (in caller order)
    assert reference_draws(seven_cubes()) == [
        (0, 0, 0, 36),    # object 1: matrix 0, the triangles of drawable 0
        (1, 0, 36, 36),   # object 3: matrix 1, drawable 1
        (0, 1, 72, 36),   # object 2: matrix 0 again, drawable 2
        (2, 1, 108, 36),  # object 6: third of its group (the unrendered one counts), drawable 3: no cursor step for object 5
        (0, 2, 144, 36),  # object 0: matrix 0, drawable 4
        (1, 2, 180, 36),  # object 4: matrix 1, drawable 5 (whose own draw was skipped)
    ]


def test_oracle_seven_cubes_known_answer(orc):
    s = seven_cubes()
    got = render_orc(orc, s, True)
    want = render_orc(orc, explicit_scene(s, reference_draws(s)), False)
    assert_frames_equal(got, want, what="seven cubes")
    assert got["stats"]["prims"] == 6 * 12
    # and it is not what the scene means: drawable 6 is never drawn, matrices 3..6 never used
    meant = render_orc(orc, s, False)
    assert not np.array_equal(meant["prim"], got["prim"])
    # the hand-made case in full: draw 3 puts drawable 3's cube (scale 0.95) where object 2 stands
    by_hand = explicit_scene(s, [(0, 0, 0, 36), (1, 0, 36, 36), (0, 1, 72, 36), (2, 1, 108, 36), (0, 2, 144, 36), (1, 2, 180, 36)])
    assert_frames_equal(got, render_orc(orc, by_hand, False), what="seven cubes by hand")


def test_oracle_mixed_index_counts(orc):
    s = hetero_scene()
    draws = reference_draws(s)
    # the case is what it claims to be: ranges that do not coincide with any drawable's own
    starts = np.concatenate([[0], np.cumsum(s.meshes["n_indices"][s.objects["mesh"]])])
    assert any(first not in starts for (_, _, first, _) in draws)
    assert (s.objects["rendered"] == 0).any() and len(set(s.objects["texture"])) == 3
    got = render_orc(orc, s, True)
    assert_frames_equal(got, render_orc(orc, explicit_scene(s, draws), False), what="mixed index counts")
    assert (got["prim"] != 0xFFFFFFFF).sum() > 2000


def test_oracle_modes_agree_without_groups(orc):
    # one texture group, everything rendered: both modes record the same draws (all benchmark scenes)
    s = scenes.config_scene(2, scale=0.08)
    s.objects["texture"] = 0
    assert_frames_equal(render_orc(orc, s, True), render_orc(orc, s, False), what="single group")
    # several groups of alike meshes, all rendered, grouped already: the index ranges are the
    # drawables' own, but the matrix index restarts with every group
    s = seven_cubes()
    s.objects["rendered"] = 1
    s.objects["texture"] = np.sort(s.objects["texture"])
    a, b = render_orc(orc, s, True), render_orc(orc, s, False)
    assert not np.array_equal(a["prim"], b["prim"])
    s.objects["texture"] = 1
    assert_frames_equal(render_orc(orc, s, True), render_orc(orc, s, False), what="one group of cubes")


def test_oracle_wants_a_texture_for_every_object(orc):
    s = seven_cubes()
    s.objects["texture"][5] = -1  # unrendered: fine by default, but it takes a place in a group under the flag
    render_orc(orc, s, False)
    with pytest.raises(ValueError):
        render_orc(orc, s, True)


# ---------------------------------------------------------------------------------------------
# GPU: the library under the flag equals the oracle under the flag
# ---------------------------------------------------------------------------------------------

def render_hip(scene, compat, use_drawables=False, shading=0):
    w, h = scene.size
    with api.HipDrawer(w, h, coverage=True, shading=shading, reference_indexing=compat) as d:
        if use_drawables:
            d.set_textures(scene.textures)
            d.update_drawables(scene.drawables())
        else:
            d.update_scene(scene)
        d.update_camera(scene.camera)
        d.draw(0)
        d.sync()
        return {"color": d.read_color(0), "depth": d.read_depth(0), "prim": d.read_coverage(0), "stats": d.stats()}


@pytest.mark.gpu
@pytest.mark.parametrize("use_drawables", [True, False], ids=["drawables", "instances"])
def test_hip_reference_indexing_equals_oracle(orc, use_drawables):
    for s in (seven_cubes(), hetero_scene(), scenes.reference_demo_scene(seed=11, width=512, height=384)):
        got = render_hip(s, True, use_drawables)
        assert_frames_equal(got, render_orc(orc, s, True), what=s.name + " (reference indexing)")
        assert got["stats"]["prims"] == sum(n for (_, _, _, n) in reference_draws(s)) // 3
        # the default mode is untouched by the flag's existence
        assert_frames_equal(render_hip(s, False, use_drawables), render_orc(orc, s, False), what=s.name)


@pytest.mark.gpu
def test_hip_modes_agree_without_groups():
    s = scenes.config_scene(2, scale=0.25)
    s.objects["texture"] = 0
    assert_frames_equal(render_hip(s, True), render_hip(s, False), what="single group")


@pytest.mark.gpu
def test_hip_reference_indexing_with_moving_matrices_and_lighting(orc):
    # update_uniforms arrives in caller order whatever the draw list does with it; the lit mode
    # reads the model matrix a draw is recorded with
    s = hetero_scene()
    w, h = s.size
    view, proj = orc.camera(s.camera, w, h)
    t = s.transforms.copy()
    with api.HipDrawer(w, h, coverage=True, shading=api.SHADE_LIT, reference_indexing=True) as d:
        d.update_scene(s)
        d.update_camera(s.camera)
        for frame in range(3):
            orc.rotation_system(t, 9, frame, 40.0)
            d.update_uniforms(orc.to_model(t))
            d.draw(frame & 1)
            d.sync()
            got = {"color": d.read_color(frame & 1), "depth": d.read_depth(frame & 1), "prim": d.read_coverage(frame & 1)}
            want = orc.render(s, orc.to_model(t), view, proj, w, h, flags=orc.FLAG_REFERENCE_INDEXING, shading=orc.SHADE_LIT)
            assert_frames_equal(got, want, what="lit, frame %d" % frame)


@pytest.mark.gpu
def test_hip_wants_a_texture_for_every_object():
    s = seven_cubes()
    s.objects["texture"][5] = 7
    with api.HipDrawer(64, 64, reference_indexing=True) as d:
        d.set_textures(s.textures)
        with pytest.raises(api.SxeError) as e:
            d.update_drawables(s.drawables())
        assert e.value.code == api.E_INVALID
