"""GPU parity: the HIP library against the CPU oracle through the C ABI (run with -m gpu)."""
import numpy as np
import pytest

from renderer_amd import api, scenes
from tests.util import assert_frames_equal, render_hip, render_oracle

pytestmark = pytest.mark.gpu


def test_cube_800x600(orc):
    s = scenes.cube_scene()
    ref = render_oracle(orc, s, want_clip=True)
    got = render_hip(s, want_clip=True)
    assert (ref["prim"] != 0xFFFFFFFF).sum() > 50000
    assert_frames_equal(got, ref, what="cfg1 cube")
    # vertex positions: bar is 1e-5 relative; the contract gives identical bits
    assert np.array_equal(got["clip"].view(np.uint32), ref["clip"].view(np.uint32))


def test_reference_demo_world(orc):
    s = scenes.reference_demo_scene()
    ref = render_oracle(orc, s)
    got = render_hip(s)
    assert_frames_equal(got, ref, what="64 cubes")
    got2 = render_hip(s, use_drawables=True)
    assert_frames_equal(got2, ref, what="64 cubes via update_drawables")


def test_sphere_field_small(orc):
    s = scenes.config_scene(2, scale=0.25)
    ref = render_oracle(orc, s)
    got = render_hip(s)
    assert_frames_equal(got, ref, what="cfg2 quarter scale")


IDENT = np.eye(4, dtype=np.float32).reshape(16)


def _soup_pair(orc, scene, **kw):
    w, h = scene.size
    models = np.tile(IDENT, (len(scene.objects), 1))
    shading = kw.pop("shading", 0)
    ref = orc.render(scene, models, IDENT, IDENT, w, h, shading=shading, **kw)
    got = render_hip(scene, models=models, view=IDENT, proj=IDENT, shading=shading, scissor=kw.get("scissor"))
    return ref, got


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_triangle_soups(orc, seed):
    """Degenerate, sub-pixel, huge, off-screen, near/far-crossing and guard-band-clipped triangles."""
    from tests.util import random_soup, soup_scene
    rng = np.random.default_rng(seed)
    w, h = [(96, 96), (200, 120), (333, 77), (64, 256)][seed - 1]  # frame sizes that are not tile multiples
    s = soup_scene(random_soup(rng, 1500, w, h), width=w, height=h, objects_split=[700, 1, 799])
    ref, got = _soup_pair(orc, s)
    assert ref["stats"]["prims_clipped"] > 50
    assert_frames_equal(got, ref, what="soup seed %d" % seed)


@pytest.mark.parametrize("shading", [0, 1, 2, 3, "blend"])
def test_vertices_and_matrices_that_are_not_finite(orc, shading):
    """NaN, infinities, values near the largest float, denormals and -0 among the positions, texture
    coordinates, colours and model matrices: such triangles are dropped or clipped the same way by
    both implementations (every comparison written so that a NaN falls on the rejecting side), and
    the rest of the frame is what it would be without them."""
    from tests.util import random_soup, soup_scene
    blend = shading == "blend"  # texture shading with a translucent texture: the ordered path
    shading = 0 if blend else shading
    rng = np.random.default_rng(35 if blend else 31 + shading)
    n = 6000
    tris = random_soup(rng, n, 256, 192)
    bad = np.array([np.nan, np.inf, -np.inf, 1e30, -1e30, 3.0e38, 1e-40, -0.0], np.float32)
    pick = rng.random((n, 3, 3)) < 0.02
    tris[pick] = rng.choice(bad, size=int(pick.sum()))
    uv = rng.uniform(0, 1, (n * 3, 2)).astype(np.float32)
    pick = rng.random((n * 3, 2)) < 0.01
    uv[pick] = rng.choice(bad, size=int(pick.sum()))
    colors = rng.uniform(0, 1, (n * 3, 3)).astype(np.float32)
    pick = rng.random((n * 3, 3)) < 0.01
    colors[pick] = rng.choice(bad, size=int(pick.sum()))
    textures = [scenes.translucent_texture(32, 9), scenes.checker_texture(16, 2)] if blend else None
    s = soup_scene(tris, uvs=uv, colors=colors, width=256, height=192, objects_split=[100] * 60, textures=textures,
                   tex_ids=(np.arange(60) % 2) if blend else None)
    w, h = s.size
    models = np.tile(IDENT, (len(s.objects), 1))
    for o, k, v in ((3, 0, np.nan), (7, 5, np.inf), (11, 15, 0.0), (13, 12, 1e30), (17, 10, -np.inf)):
        models[o, k] = v
    ref = orc.render(s, models, IDENT, IDENT, w, h, shading=shading)
    got = render_hip(s, models=models, view=IDENT, proj=IDENT, shading=shading)
    assert (ref["prim"] != 0xFFFFFFFF).mean() > 0.5
    # (a colour may be NaN where a NaN attribute was interpolated: compared as bits like everything else)
    assert_frames_equal(got, ref, what="non-finite input")


def test_camera_and_transforms_that_are_not_finite(orc):
    """A camera or an object pose with NaN or infinite components (update_camera and
    update_transforms take whatever the ECS holds): the affected objects draw nothing or the same
    pixels as in the oracle, the others are untouched, and the rotation system keeps going."""
    s = scenes.config_scene(2, scale=0.2)
    w, h = s.size
    for k, (field, idx, value) in enumerate([("position", 0, np.nan), ("rotation", 1, np.inf), ("scale", 2, -np.inf), (None, 0, 0)]):
        cam = s.camera.copy()
        if field:
            cam[field][0][idx] = value
        t = s.transforms.copy()
        t["position"][1][0] = np.nan
        t["scale"][2][1] = np.inf
        t["rotation"][3][2] = np.nan
        t["scale"][4][:] = 0.0
        with api.HipDrawer(w, h, coverage=True) as d:
            d.update_scene(s)
            d.update_transforms(t)
            d.update_camera(cam)
            d.rotation_system(5, k, 16.0)
            d.draw(0)
            d.sync()
            t_after = d.read_transforms()
            got = {"color": d.read_color(0), "depth": d.read_depth(0), "prim": d.read_coverage(0)}
        t_ref = t.copy()
        orc.rotation_system(t_ref, 5, k, 16.0)
        # (bit for bit, except that a NaN may carry another sign or payload on the other processor)
        fa, fb = t_after.view(np.float32), t_ref.view(np.float32)
        assert ((fa.view(np.uint32) == fb.view(np.uint32)) | (np.isnan(fa) & np.isnan(fb))).all()
        ref = orc.render(s, orc.to_model(t_ref), *orc.camera(cam, w, h), w, h)
        assert_frames_equal(got, ref, what="non-finite %s" % (field or "poses only"))
        if not field:
            assert (ref["prim"] != 0xFFFFFFFF).any()


def test_fan_triangles_that_overlap_by_a_sliver(orc):
    """A triangle cut by the near and the far plane: the corners of its fan are snapped to the grid
    one by one, and at pixel (33, 51) of this frame two of the fan's triangles both cover the
    sample.  Drawn in turn under Less, the one with the smaller depth stays, and its attribute
    planes, not those of the first one that covers, must shade the pixel (found by
    tools/fuzz_parity.py: one pixel of one scene in twelve thousand)."""
    from tests.util import soup_scene
    tri = np.array([[[0.55813956, -0.48756218, -0.02677643], [0.5810299, -0.33630487, 0.6650742],
                     [0.49605954, -0.896145, 1.0457937]]], dtype=np.float32)
    uv = np.array([[0.04350607, -0.09129993], [0.7519953, 1.4499753], [-0.29209238, 0.08491877]], dtype=np.float32)
    rng = np.random.default_rng(5)
    colors = rng.uniform(0, 1, (3, 3))
    tex = rng.integers(0, 256, size=(17, 23, 4), dtype=np.uint8)
    for translucent in (False, True):
        t = tex.copy()
        if not translucent:
            t[..., 3] = 255
        s = soup_scene(tri, uvs=uv, colors=colors, textures=[t], width=43, height=201)
        for shading in (0, 1, 2, 3):
            ref, got = _soup_pair(orc, s, shading=shading)
            assert ref["stats"]["prims_clipped"] == 1 and ref["prim"][51, 33] == 0
            assert_frames_equal(got, ref, what="overlapping fan triangles, shading %d%s" % (shading, ", translucent" if translucent else ""))
    # the same sliver among a crowd of clipped triangles: random triangles through both planes
    n = 3000
    tris = np.zeros((n, 3, 3), np.float32)
    c = rng.uniform(-0.9, 0.9, (n, 1, 2))
    tris[..., :2] = c + rng.uniform(-0.5, 0.5, (n, 3, 2))
    tris[..., 2] = rng.uniform(-0.6, 1.6, (n, 3))
    s = soup_scene(tris, uvs=rng.uniform(-0.5, 1.5, (3 * n, 2)), colors=rng.uniform(0, 1, (3 * n, 3)), textures=[tex | np.uint8(0)], width=211, height=173)
    s.textures[0][..., 3] = 255
    ref, got = _soup_pair(orc, s, shading=2)
    assert ref["stats"]["prims_clipped"] > 1000
    assert_frames_equal(got, ref, what="crowd of triangles through both planes")


def test_more_textures_than_the_tile_kernel_keeps_in_lds(orc):
    """The tile kernel keeps the descriptors of up to 64 textures in LDS; a longer table is read
    from memory (a build of the kernel of its own).  Ninety objects with a texture each, opaque
    and with some translucent ones, in every shading mode that samples a texture."""
    from tests.util import random_soup, soup_scene
    rng = np.random.default_rng(64)
    n_obj, per = 90, 12
    texs = [rng.integers(0, 256, size=(int(rng.integers(1, 20)), int(rng.integers(1, 20)), 4), dtype=np.uint8) for _ in range(n_obj)]
    for translucent in (False, True):
        for k, t in enumerate(texs):
            t[..., 3] = rng.integers(0, 256, size=t.shape[:2], dtype=np.uint8) if (translucent and k % 7 == 0) else 255
        n = n_obj * per
        s = soup_scene(random_soup(rng, n, 150, 110, kinds=("small", "medium", "large", "near")), uvs=rng.uniform(-0.2, 1.2, (3 * n, 2)),
                       colors=rng.uniform(0, 1, (3 * n, 3)), textures=texs, width=150, height=110, objects_split=[per] * n_obj,
                       tex_ids=rng.permutation(n_obj))
        for shading in (0, 2, 3):
            ref, got = _soup_pair(orc, s, shading=shading)
            assert_frames_equal(got, ref, what="90 textures, shading %d%s" % (shading, ", some translucent" if translucent else ""))


def test_micro_triangles_dense(orc):
    """Many sub-pixel triangles in few tiles: stresses the bin allocation and depth ties."""
    from tests.util import soup_scene
    rng = np.random.default_rng(77)
    n = 60000
    c = rng.uniform(-0.3, 0.3, (n, 1, 2))
    tris = np.zeros((n, 3, 3), np.float32)
    tris[..., :2] = c + rng.uniform(-0.02, 0.02, (n, 3, 2))
    tris[..., 2] = np.round(rng.uniform(0.1, 0.9, (n, 1)) * 8) / 8  # few distinct depths -> many exact ties
    s = soup_scene(tris, width=128, height=128)
    ref, got = _soup_pair(orc, s)
    assert_frames_equal(got, ref, what="dense micro triangles")


@pytest.mark.parametrize("shading", [1, 2])
def test_shading_modes(orc, shading):
    """standard.frag:13 (vertex colour) and ui.frag:12 (colour x texture)."""
    s = scenes.config_scene(2, scale=0.15)
    ref = render_oracle(orc, s, shading=shading)
    got = render_hip(s, shading=shading)
    assert_frames_equal(got, ref, what="shading %d" % shading)
    from tests.util import random_soup, soup_scene
    rng = np.random.default_rng(5)
    tris = random_soup(rng, 600, 128, 128)
    s = soup_scene(tris, colors=rng.uniform(0, 1, (600 * 3, 3)), width=128, height=128)
    ref, got = _soup_pair(orc, s, shading=shading)
    assert_frames_equal(got, ref, what="soup shading %d" % shading)


def test_lit_shading(orc):
    """SXE_SHADE_LIT, the flat Blinn-Phong extension (no reference counterpart: parity is with the
    oracle's restatement of the definition in include/sxe.h): cube, sphere field under two lights,
    clipped soup, translucent textures, a set of views, and the light's validation."""
    from tests.util import random_soup, soup_scene
    lights = [None, dict(to_light=(-1.0, 0.2, 0.3), to_viewer=(0.2, 0.1, 1.0), ambient=0.05, diffuse=0.9, specular=0.6, shininess_log2=3),
              dict(to_light=(0.0, -1.0, 0.0), specular=0.0, shininess_log2=0)]
    c = scenes.cube_scene(320, 240)
    s = scenes.config_scene(2, scale=0.3)
    for light in lights:
        for scene in (c, s):
            ref = render_oracle(orc, scene, shading=api.SHADE_LIT, light=light)
            got = render_hip(scene, shading=api.SHADE_LIT, light=light)
            assert_frames_equal(got, ref, what="lit %s, light %r" % (scene.name, light))
    plain = render_oracle(orc, s)
    assert (plain["prim"] == ref["prim"]).all() and not np.array_equal(plain["color"], ref["color"])
    # triangles cut by the near plane and the guard band keep their face's terms
    rng = np.random.default_rng(12)
    w, h = 200, 120
    soup = soup_scene(random_soup(rng, 900, w, h), width=w, height=h)
    ref, got = _soup_pair(orc, soup, shading=api.SHADE_LIT)
    assert ref["stats"]["prims_clipped"] > 30
    assert_frames_equal(got, ref, what="lit soup")
    # lit and blended
    t = scenes.config_scene(2, scale=0.3)
    t.textures = [t.textures[0], scenes.translucent_texture(128, 5), t.textures[2], scenes.translucent_texture(64, 6)]
    ref = render_oracle(orc, t, shading=api.SHADE_LIT, light=lights[1])
    got = render_hip(t, shading=api.SHADE_LIT, light=lights[1])
    assert_frames_equal(got, ref, what="lit, two of four textures translucent")
    # a set of views shares the light
    v = scenes.sphere_field_scene(20, 4, 5.0, (-14.0, -4.0), 160, 96, n_textures=2, tex_size=64, name="views")
    cams = scenes.orbit_cameras(4, 9, radius=12.0, target=(0.0, 0.0, -9.0))
    models = orc.to_model(v.transforms)
    with api.HipDrawer(160, 96, n_images=4, coverage=True, shading=api.SHADE_LIT) as d:
        d.update_scene(v)
        d.set_light(**lights[1])
        d.draw_views(cams)
        for k in range(4):
            view, proj = orc.camera(cams[k:k + 1], 160, 96)
            ref = orc.render(v, models, view, proj, 160, 96, shading=api.SHADE_LIT, light=lights[1])
            got = {"color": d.read_color(k), "depth": d.read_depth(k), "prim": d.read_coverage(k)}
            assert_frames_equal(got, ref, what="lit view %d" % k)
        for bad in (dict(to_light=(0, 0, 0)), dict(to_viewer=(float("inf"), 0, 0)), dict(shininess_log2=17), dict(ambient=float("nan"))):
            with pytest.raises(api.SxeError) as e:
                d.set_light(**bad)
            assert e.value.code == api.E_INVALID


def test_tiny_textures_and_coordinates_beyond_the_edges(orc):
    """Clamp to edge (the sampler of src/renderer/drawer.rs:231) where it matters most: textures of
    one texel, one row or one column, odd sizes, and texture coordinates far outside [0, 1], so that
    samples land on and beyond every border of the texture."""
    from tests.util import random_soup, soup_scene
    rng = np.random.default_rng(77)
    sizes = [(1, 1), (1, 7), (5, 1), (3, 2), (2, 2), (17, 9)]  # (width, height)
    textures = [rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8) for (w, h) in sizes]
    for t in textures:
        t[..., 3] = 255
    n_obj, per = len(sizes), 150
    tris = random_soup(rng, n_obj * per, 128, 128, kinds=("small", "medium", "large", "huge"))
    uvs = rng.uniform(-0.75, 1.75, size=(n_obj * per * 3, 2)).astype(np.float32)
    uvs[::7] = rng.choice(np.array([0.0, 1.0, -0.0, 0.5], np.float32), size=uvs[::7].shape)  # exactly on the borders too
    s = soup_scene(tris, uvs=uvs, textures=textures, width=128, height=128, objects_split=[per] * n_obj,
                   tex_ids=np.arange(n_obj))
    ref, got = _soup_pair(orc, s)
    assert_frames_equal(got, ref, what="tiny textures")


def test_textures_of_the_largest_extent(orc):
    """Rows and columns of 16384 texels, the most a texture may have: footprint addressing at the far
    end of a table (24-bit row multiply), the clamp there, and the refusal of one texel more."""
    from tests.util import random_soup, soup_scene
    rng = np.random.default_rng(78)
    sizes = [(16384, 1), (1, 16384), (16384, 3), (2, 16384)]
    textures = [rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8) for (w, h) in sizes]
    for t in textures:
        t[..., 3] = 255
    n_obj, per = len(sizes), 120
    tris = random_soup(rng, n_obj * per, 160, 120, kinds=("small", "medium", "large", "huge"))
    uvs = rng.uniform(-0.1, 1.1, size=(n_obj * per * 3, 2)).astype(np.float32)
    uvs[::5] = rng.choice(np.array([0.0, 1.0, 0.99999, 0.5], np.float32), size=uvs[::5].shape)
    s = soup_scene(tris, uvs=uvs, textures=textures, width=160, height=120, objects_split=[per] * n_obj,
                   tex_ids=np.arange(n_obj))
    ref, got = _soup_pair(orc, s)
    assert_frames_equal(got, ref, what="largest textures")
    with api.HipDrawer(64, 64) as d:
        with pytest.raises(api.SxeError):
            d.set_textures([np.full((1, 16385, 4), 255, np.uint8)])


def test_random_scenes(orc):
    """A hundred random scenes (tools/fuzz_parity.py: soups of every awkward kind at odd frame sizes,
    random shading mode, scissor band, object split, tiny textures, some translucent; sphere fields
    under random cameras): colour, depth and primitive id equal the oracle's in every one."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                              "tools", "fuzz_parity.py"))
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    assert fuzz.run(100, 3, verbose=True) == 0


def test_scissor_bands_equal_the_full_frame(orc):
    s = scenes.config_scene(2, scale=0.3)
    w, h = s.size
    full = render_oracle(orc, s)
    from renderer_amd import shard
    for world in (2, 3):
        for r in range(world):
            y0, y1 = shard.band_rows(h, world, r)
            got = render_hip(s, scissor=(y0, y1))
            assert_frames_equal(got, full, rows=(y0, y1), what="band %d/%d" % (r, world))
    got = render_hip(s, scissor=(37, 101))  # not tile aligned
    assert_frames_equal(got, full, rows=(37, 101), what="odd band")


def test_device_transform_kernels(orc):
    """Transform::to_ubo and the rotation system on the device against the oracle, bit for bit."""
    s = scenes.config_scene(3, scale=0.03)
    w, h = s.size
    with api.HipDrawer(w, h, coverage=True) as d:
        d.update_scene(s)
        assert d.read_models().tobytes() == orc.to_model(s.transforms).tobytes()
        t = s.transforms.copy()
        t["position"] += 1.0
        d.update_transforms(t)
        assert d.read_models().tobytes() == orc.to_model(t).tobytes()
        for f in range(5):
            d.rotation_system(3, f, 16.0)
            orc.rotation_system(t, 3, f, 16.0)
        assert d.read_transforms().tobytes() == t.tobytes()
        assert d.read_models().tobytes() == orc.to_model(t).tobytes()
        # seven more updates in one launch equal seven launches
        d.rotation_system(3, 5, 16.0, n_frames=7)
        for f in range(5, 12):
            orc.rotation_system(t, 3, f, 16.0)
        assert d.read_transforms().tobytes() == t.tobytes()
        assert d.read_models().tobytes() == orc.to_model(t).tobytes()
        d.update_camera(s.camera)
        d.draw(1)
        got = {"color": d.read_color(1), "depth": d.read_depth(1), "prim": d.read_coverage(1)}
        ref = orc.render(s, orc.to_model(t), *orc.camera(s.camera, w, h), w, h)
        assert_frames_equal(got, ref, what="after 12 rotation steps")
        # an update that is followed by a draw is carried out by the frame's first kernel: the
        # transforms and matrices in memory must come out the same
        d.rotation_system(3, 12, 16.0)
        d.draw(0)
        orc.rotation_system(t, 3, 12, 16.0)
        assert d.read_transforms().tobytes() == t.tobytes()
        assert d.read_models().tobytes() == orc.to_model(t).tobytes()
        got = {"color": d.read_color(0), "depth": d.read_depth(0), "prim": d.read_coverage(0)}
    ref = orc.render(s, orc.to_model(t), *orc.camera(s.camera, w, h), w, h)
    assert_frames_equal(got, ref, what="after 13 rotation steps")


def test_moving_camera(orc):
    s = scenes.config_scene(2, scale=0.25)
    w, h = s.size
    cams = scenes.orbit_cameras(3, 12, radius=30.0, target=(0.0, 0.0, -22.5))
    with api.HipDrawer(w, h, coverage=True) as d:
        d.update_scene(s)
        for i in range(len(cams)):
            d.update_camera(cams[i:i + 1])
            d.draw(i & 1)
            got = {"color": d.read_color(i & 1), "depth": d.read_depth(i & 1), "prim": d.read_coverage(i & 1)}
            view, proj = orc.camera(cams[i:i + 1], w, h)
            ref = orc.render(s, orc.to_model(s.transforms), view, proj, w, h)
            assert (ref["prim"] != 0xFFFFFFFF).sum() > 100
            assert_frames_equal(got, ref, what="camera %d" % i)


def test_clip_positions_and_uniform_upload(orc):
    s = scenes.config_scene(2, scale=0.1)
    w, h = s.size
    rng = np.random.default_rng(2)
    models = orc.to_model(s.transforms)
    models += rng.normal(0, 1e-3, models.shape).astype(np.float32)  # arbitrary matrices via update_uniforms
    view, proj = orc.camera(s.camera, w, h)
    ref = orc.render(s, models, view, proj, w, h, want_clip=True)
    got = render_hip(s, models=models, want_clip=True)
    assert_frames_equal(got, ref, what="update_uniforms")
    rel = np.abs(got["clip"] - ref["clip"]) / np.maximum(np.abs(ref["clip"]), 1e-6)
    assert rel.max() <= 1e-5  # the stated bar
    assert np.array_equal(got["clip"].view(np.uint32), ref["clip"].view(np.uint32))  # what the contract gives


def test_work_buffers_grow_on_demand(orc):
    """A frame that needs more record / bin space than the first guess is re-rendered after growth."""
    from tests.util import soup_scene
    rng = np.random.default_rng(8)
    n = 3000
    tris = np.zeros((n, 3, 3), np.float32)
    tris[..., :2] = rng.uniform(-1, 1, (n, 3, 2))  # every triangle spans many tiles
    tris[..., 2] = rng.uniform(0.1, 0.9, (n, 1))
    s = soup_scene(tris, width=512, height=512)
    models = np.tile(IDENT, (1, 1))
    ref = orc.render(s, models, IDENT, IDENT, 512, 512)
    got = render_hip(s, models=models, view=IDENT, proj=IDENT)
    assert got["stats"]["redraws"] >= 1 and got["stats"]["bin_entries"] > 4096 + n // 2 + 4 * 256
    assert_frames_equal(got, ref, what="regrown buffers")


def test_config5_views_share_one_scene(orc):
    """The multi-view configuration: one resident scene, a new camera per 512x512 view, frames in
    flight; the views a rank of N would render (shard.views_for_rank) are checked one by one."""
    from renderer_amd import shard
    s = scenes.config_scene(5, scale=0.5)
    w, h = s.size
    n_views, world = 12, 4
    cams = scenes.orbit_cameras(n_views, 5, radius=30.0, target=(0.0, 0.0, -22.5))
    models = orc.to_model(s.transforms)
    for rank in (0, 3):
        mine = list(range(*shard.views_for_rank(n_views, world, rank)))
        assert len(mine) == 3
        with api.HipDrawer(w, h, n_images=len(mine), coverage=True) as d:
            d.update_scene(s)
            for k, v in enumerate(mine):
                d.update_camera(cams[v:v + 1])
                d.draw(k)
            d.sync()
            for k, v in enumerate(mine):
                view, proj = orc.camera(cams[v:v + 1], w, h)
                ref = orc.render(s, models, view, proj, w, h)
                got = {"color": d.read_color(k), "depth": d.read_depth(k), "prim": d.read_coverage(k)}
                assert_frames_equal(got, ref, what="view %d (rank %d)" % (v, rank))


def test_view_pool_renders_every_view_like_a_single_drawer(orc):
    """Views dealt out over three contexts with two images each: images are reused while later
    views are in flight, and every view must still equal the oracle's."""
    from renderer_amd import shard
    s = scenes.config_scene(5, scale=0.5)
    w, h = s.size
    cams = scenes.orbit_cameras(11, 5, radius=30.0, target=(0.0, 0.0, -22.5))
    models = orc.to_model(s.transforms)
    with shard.ViewPool(s, contexts=3, n_images=2, coverage=True) as pool:
        got = pool.render(cams, read=lambda d, img: {"color": d.read_color(img), "depth": d.read_depth(img), "prim": d.read_coverage(img)})
    assert len(got) == 11
    for v in range(11):
        view, proj = orc.camera(cams[v:v + 1], w, h)
        assert_frames_equal(got[v], orc.render(s, models, view, proj, w, h), what="pooled view %d" % v)


def test_config4_bands_of_an_8k_aspect_frame(orc):
    """The sort-first configuration at reduced size: 8 bands of whole tile rows of a 16:9 frame whose
    height is not a multiple of the tile size reassemble to the full frame."""
    from renderer_amd import shard
    s = scenes.config_scene(4, scale=0.05)  # 384 x 216, 500 objects
    w, h = s.size
    ref = orc.render(s, orc.to_model(s.transforms), *orc.camera(s.camera, w, h), w, h)
    color = np.zeros((h, w, 4), np.uint8)
    depth = np.zeros((h, w), np.float32)
    covered = np.zeros(h, bool)
    for rank in range(8):
        y0, y1 = shard.band_rows(h, 8, rank)
        if y1 <= y0:
            continue
        got = render_hip(s, scissor=(y0, y1))
        color[y0:y1] = got["color"][y0:y1]
        depth[y0:y1] = got["depth"][y0:y1]
        assert not covered[y0:y1].any()
        covered[y0:y1] = True
    assert covered.all()
    assert np.array_equal(color, ref["color"]) and np.array_equal(depth.view(np.uint32), ref["depth"].view(np.uint32))


def test_frame_with_more_than_8192_tiles_and_crowds_of_large_triangles(orc):
    """4096 x 2112 is 128 x 66 = 8448 tiles: bin offsets and launch order are worked out by several
    workgroups.  Spheres close to the camera put hundreds of triangles that span more than four
    tiles on the list of late bin entries; some of them cross the near plane and arrive as fans."""
    s = scenes.sphere_field_scene(40, 11, 6.0, (-14.0, -0.6), 4096, 2112, n_textures=2, tex_size=64, name="near-spheres")
    w, h = s.size
    ref = orc.render(s, orc.to_model(s.transforms), *orc.camera(s.camera, w, h), w, h, threads=8)
    with api.HipDrawer(w, h, coverage=True) as d:
        d.update_scene(s)
        d.update_camera(s.camera)
        d.draw(0)
        d.sync()
        assert d.stats()["tiles"] == 8448
        parts = d.tile_parts()
        assert parts[:, -1].sum() > 2000, "the scene is meant to produce many late entries"
        got = {"color": d.read_color(0), "depth": d.read_depth(0), "prim": d.read_coverage(0)}
    assert_frames_equal(got, ref, what="8448 tiles")


def test_many_small_objects(orc):
    """150 000 objects of one triangle each (the reference's loop has one draw per mesh,
    src/renderer/drawer.rs:313-347): per-object arrays, work items and visibility flags far beyond
    the benchmark's 10 000, every work item 255/256 empty; drawn twice with a rotation in between."""
    from tests.util import random_soup, soup_scene
    rng = np.random.default_rng(5)
    n = 150000
    tris = random_soup(rng, n, 640, 480, kinds=("small", "medium"))
    s = soup_scene(tris, width=640, height=480, objects_split=[1] * n)
    w, h = s.size
    view, proj = orc.camera(s.camera, w, h)
    with api.HipDrawer(w, h, coverage=True) as d:
        d.update_scene(s)
        d.update_camera(s.camera)
        for frame in range(2):
            d.rotation_system(9, frame, 16.0)
            d.draw(frame & 1)
        d.sync()
        assert d.stats()["objects"] == n
        t = d.read_transforms()
        got = {"color": d.read_color(1), "depth": d.read_depth(1), "prim": d.read_coverage(1)}
    ref = orc.render(s, orc.to_model(t), view, proj, w, h, threads=16)
    assert_frames_equal(got, ref, what="150 000 objects")


def test_the_largest_frame(orc):
    """8192 x 8192, the stated limit: 65 536 tiles, pixel addresses up to 2^26, and one pixel more
    either way is refused."""
    s = scenes.sphere_field_scene(60, 12, 8.0, (-20.0, -1.0), api.MAX_DIM, api.MAX_DIM, n_textures=2, tex_size=32, name="largest-frame")
    w, h = s.size
    ref = orc.render(s, orc.to_model(s.transforms), *orc.camera(s.camera, w, h), w, h, threads=16)
    with api.HipDrawer(w, h, n_images=1, coverage=True) as d:
        d.update_scene(s)
        d.update_camera(s.camera)
        d.draw(0)
        d.sync()
        assert d.stats()["tiles"] == 256 * 256
        got = {"color": d.read_color(0), "depth": d.read_depth(0), "prim": d.read_coverage(0)}
    assert_frames_equal(got, ref, what="8192 x 8192")
    for bad in ((api.MAX_DIM + 1, 64), (64, api.MAX_DIM + 1)):
        with pytest.raises(api.SxeError):
            api.HipDrawer(*bad)


def test_frames_in_flight(orc):
    """Frames enqueued back to back without a host wait: every image must hold exactly its frame,
    also when the first frames overflow their work buffers and are drawn again after the fact
    (with the matrices they were submitted with, although the transforms have moved on)."""
    s = scenes.config_scene(3, scale=0.03)
    w, h = s.size
    view, proj = orc.camera(s.camera, w, h)
    for n_images in (1, 3):
        t = s.transforms.copy()
        refs = []
        with api.HipDrawer(w, h, n_images=n_images, coverage=True, timing=True) as d:
            d.update_scene(s)
            d.update_camera(s.camera)
            for f in range(5):
                d.rotation_system(3, f, 400.0)  # large steps: consecutive frames differ visibly
                orc.rotation_system(t, 3, f, 400.0)
                d.draw(f % n_images)
                refs.append(orc.render(s, orc.to_model(t), view, proj, w, h))
            d.sync()
            sums, frames = d.timing_totals()
            assert frames == 5 and sums["raster_shade_ms"] > 0.0
            # events on a sample of the frames only: frames 6 and 9 of the ten drawn so far
            d.set_timing_stride(3)
            d.reset_timing_totals()
            for f in range(5, 10):
                d.rotation_system(3, f, 400.0)
                orc.rotation_system(t, 3, f, 400.0)
                d.draw(f % n_images)
                refs.append(orc.render(s, orc.to_model(t), view, proj, w, h))
            d.sync()
            sums, frames = d.timing_totals()
            assert frames == 2 and sums["raster_shade_ms"] > 0.0 and sums["object_ms"] > 0.0
            # a stride of 0 switches the events off: frames are drawn, none is timed
            d.set_timing_stride(0)
            d.reset_timing_totals()
            d.draw(9 % n_images)
            d.sync()
            assert d.timing_totals()[1] == 0
            for img in range(n_images):
                last = max(f for f in range(10) if f % n_images == img)
                got = {"color": d.read_color(img), "depth": d.read_depth(img), "prim": d.read_coverage(img)}
                assert_frames_equal(got, refs[last], what="image %d of %d holds frame %d" % (img, n_images, last))
        assert not np.array_equal(refs[3]["color"], refs[4]["color"])


def test_frames_in_flight_with_late_overflow(orc):
    """The first frames of a scene whose bins outgrow the first guess are in flight together."""
    from tests.util import soup_scene
    rng = np.random.default_rng(9)
    n = 3000
    tris = np.zeros((n, 3, 3), np.float32)
    tris[..., :2] = rng.uniform(-1, 1, (n, 3, 2))
    tris[..., 2] = rng.uniform(0.1, 0.9, (n, 1))
    s = soup_scene(tris, width=512, height=512)
    m0 = np.tile(IDENT, (1, 1)).astype(np.float32)
    m1 = m0.copy()
    m1[0, 12] = 0.25  # second frame: the whole soup moved to the right
    refs = [orc.render(s, m, IDENT, IDENT, 512, 512) for m in (m0, m1)]
    with api.HipDrawer(512, 512, n_images=2, coverage=True) as d:
        d.update_scene(s)
        d.update_camera_matrices(IDENT, IDENT)
        d.update_uniforms(m0)
        d.draw(0)
        d.update_uniforms(m1)
        d.draw(1)
        d.sync()
        assert d.stats()["redraws"] >= 2
        for img in (0, 1):
            got = {"color": d.read_color(img), "depth": d.read_depth(img), "prim": d.read_coverage(img)}
            assert_frames_equal(got, refs[img], what="late overflow, image %d" % img)


def test_lit_frames_in_flight_with_late_overflow_keep_their_own_light_and_matrices(orc):
    """A lit frame that overflows is drawn again when it is retired, after younger frames have
    rotated the models and the light has changed: it must be lit as it was when it was drawn
    (matrices and light are kept with the frame, like its projection matrices)."""
    rng = np.random.default_rng(10)
    s = scenes.sphere_field_scene(40, 31, 2.5, (-9.0, -4.0), 512, 384, n_textures=2, tex_size=64)
    w, h = s.size
    view, proj = orc.camera(s.camera, w, h)
    lights = [dict(to_light=(0.3, 0.8, 0.5), ambient=0.2), dict(to_light=(-0.7, 0.1, 0.6), ambient=0.35, specular=0.6)]
    t = s.transforms.copy()
    refs = []
    with api.HipDrawer(w, h, n_images=2, coverage=True, shading=api.SHADE_LIT) as d:
        d.update_scene(s)
        d.update_camera(s.camera)
        for f in range(2):  # both in flight before the first one is looked at
            d.set_light(**lights[f])
            d.rotation_system(5, f, 900.0)
            orc.rotation_system(t, 5, f, 900.0)
            d.draw(f)
            refs.append(orc.render(s, orc.to_model(t), view, proj, w, h, shading=orc.SHADE_LIT, light=lights[f]))
        d.sync()
        assert d.stats()["redraws"] >= 2  # (40 spheres of 1024 triangles close to the camera outgrow the first bins)
        for img in (0, 1):
            got = {"color": d.read_color(img), "depth": d.read_depth(img), "prim": d.read_coverage(img)}
            assert_frames_equal(got, refs[img], what="lit, late overflow, image %d" % img)
    assert not np.array_equal(refs[0]["color"], refs[1]["color"])
    del rng


def test_clipping_that_appears_after_quiet_frames(orc):
    """The clip kernel is left out after some frames without a triangle to clip; the first frame
    that needs it again must still come out right (it is drawn again), also with frames in flight."""
    s = scenes.cube_scene(160, 120)
    w, h = s.size
    view, proj = orc.camera(s.camera, w, h)
    far = s.transforms.copy()
    near = s.transforms.copy()
    near["position"][0] = (0.1, 0.0, -0.45)  # the cube now straddles the near plane
    with api.HipDrawer(w, h, n_images=2, coverage=True) as d:
        d.update_scene(s)
        d.update_camera(s.camera)
        for f in range(12):
            d.draw(f & 1)
        d.sync()
        assert d.stats()["redraws"] == 0
        d.update_transforms(near)
        d.draw(0)
        d.update_transforms(far)
        d.draw(1)
        d.update_transforms(near)
        d.draw(0)
        d.sync()
        assert d.stats()["redraws"] >= 1
        got_near = {"color": d.read_color(0), "depth": d.read_depth(0), "prim": d.read_coverage(0)}
        got_far = {"color": d.read_color(1), "depth": d.read_depth(1), "prim": d.read_coverage(1)}
    ref_near = orc.render(s, orc.to_model(near), view, proj, w, h)
    ref_far = orc.render(s, orc.to_model(far), view, proj, w, h)
    assert not np.array_equal(ref_near["color"], ref_far["color"])
    assert_frames_equal(got_near, ref_near, what="cube across the near plane")
    assert_frames_equal(got_far, ref_far, what="cube back in front")


def test_contexts_give_back_what_they_took():
    """Device memory after a hundred contexts have been created, used (uploads, frames whose work
    buffers grow, a view set, the allocation-free draw, readbacks) and destroyed is what it was
    after the first one."""
    import torch
    s = scenes.config_scene(2, scale=0.25)
    w, h = s.size
    cams = scenes.orbit_cameras(3, 5)

    def once():
        with api.HipDrawer(w, h, n_images=3, coverage=True, timing=True) as d:
            d.update_scene(s)
            d.update_camera(s.camera)
            for i in range(4):
                d.rotation_system(1, i, 16.0)
                d.draw(i % 3)
            d.sync()
            d.draw_views(cams, 0)
            d.sync()
            d.draw_noalloc(1)
            d.read_color(1)
            d.map_color(0)

    once()
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    for _ in range(100):
        once()
    torch.cuda.synchronize()
    free_after = torch.cuda.mem_get_info()[0]
    assert free_before - free_after < (8 << 20), "device memory shrank by %d bytes over 100 contexts" % (free_before - free_after)


def test_mapped_frame_equals_the_copied_one():
    s = scenes.cube_scene(160, 120)
    with api.HipDrawer(160, 120, n_images=2) as d:
        with pytest.raises(api.SxeError):
            d.map_color(0)  # nothing drawn yet
        d.update_scene(s)
        d.update_camera(s.camera)
        d.draw(0)
        d.rotation_system(3, 0, 400.0)
        d.draw(1)
        first = d.map_color(0)
        assert first.shape == (120, 160, 4) and not first.flags.writeable
        assert np.array_equal(first, d.read_color(0))
        second = d.map_color(1)
        assert np.array_equal(second, d.read_color(1)) and not np.array_equal(first, second)
        assert np.array_equal(first, d.read_color(0))  # mapping image 1 left image 0's copy alone


def test_error_behaviour(orc):
    s = scenes.cube_scene(64, 64)
    with api.HipDrawer(64, 64) as d:
        with pytest.raises(api.SxeError) as e:
            d.draw(0)
        assert e.value.code == api.E_STATE
        with pytest.raises(api.SxeError) as e:
            d.update_uniforms(np.zeros((1, 16), np.float32))
        assert e.value.code == api.E_STATE
        d.set_textures([])
        with pytest.raises(api.SxeError) as e:  # drawable without a texture (the reference panics: drawer.rs:317)
            d.update_drawables(s.drawables())
        assert e.value.code == api.E_INVALID and "texture" in str(e.value)
        d.update_scene(s)
        with pytest.raises(api.SxeError) as e:
            d.draw(0)  # no camera yet
        assert e.value.code == api.E_STATE
        d.update_camera(s.camera)
        with pytest.raises(api.SxeError) as e:
            d.draw(5)
        assert e.value.code == api.E_INVALID
        with pytest.raises(api.SxeError) as e:  # more matrices than objects (types.rs:51 asserts instead)
            d.update_uniforms(np.zeros((2, 16), np.float32))
        assert e.value.code == api.E_CAPACITY
        with pytest.raises(api.SxeError) as e:
            d.read_color(1)  # never drawn
        assert e.value.code == api.E_STATE
        with pytest.raises(api.SxeError):
            d.read_coverage(0)  # drawn? no: and coverage was not requested
        d.draw(0)
        d.sync()
        bad = s.drawables()
        bad[0].indices = bad[0].indices.copy()
        bad[0].indices[5] = 24
        with pytest.raises(api.SxeError) as e:
            d.update_drawables(bad)
        assert e.value.code == api.E_INVALID and "out of range" in str(e.value)
        # the context is still usable after errors
        d.update_scene(s)
        d.draw(0)
        assert (d.read_depth(0) < 1.0).sum() > 100


def test_rerender_is_idempotent_and_images_are_independent(orc):
    s = scenes.config_scene(2, scale=0.2)
    w, h = s.size
    with api.HipDrawer(w, h, n_images=3, coverage=True) as d:
        d.update_scene(s)
        d.update_camera(s.camera)
        d.draw(0)
        a = d.read_color(0)
        t = s.transforms.copy()
        t["position"][:, 0] += 0.5
        d.update_transforms(t)
        d.draw(1)
        d.update_transforms(s.transforms)
        d.draw(2)
        assert np.array_equal(d.read_color(0), a) and np.array_equal(d.read_color(2), a)
        assert not np.array_equal(d.read_color(1), a)


def test_full_size_config3_against_oracle(orc):
    """The benchmark workload itself: 10.24 M triangles at 3840x2160, compared with the
    band-parallel oracle, plus size-independent properties (idempotence, band union)."""
    s = scenes.config_scene(3)
    w, h = s.size
    with api.HipDrawer(w, h, coverage=True) as d:
        d.update_scene(s)
        d.update_camera(s.camera)
        d.rotation_system(3, 0, 16.0)
        d.draw(0)
        got = {"color": d.read_color(0), "depth": d.read_depth(0), "prim": d.read_coverage(0)}
        t = d.read_transforms()
        d.draw(1)
        assert np.array_equal(d.read_color(1), got["color"]) and np.array_equal(d.read_depth(1), got["depth"])
        st = d.stats()
    assert st["prims"] == 10240000
    ref = orc.render(s, orc.to_model(t), *orc.camera(s.camera, w, h), w, h, threads=16)
    assert_frames_equal(got, ref, what="config 3 full size")
    from renderer_amd import shard
    y0, y1 = shard.band_rows(h, 4, 2)
    with api.HipDrawer(w, h, coverage=True, scissor=(y0, y1)) as d:
        d.update_scene(s, transforms=t)
        d.update_camera(s.camera)
        d.draw(0)
        band = {"color": d.read_color(0), "depth": d.read_depth(0), "prim": d.read_coverage(0)}
    assert_frames_equal(band, ref, rows=(y0, y1), what="config 3 band 2/4")


# ---- the benchmark configurations at their full sizes ----

def test_full_size_config2_against_oracle(orc):
    """Config 2: 100 spheres = 102,400 triangles at 1920x1080."""
    s = scenes.config_scene(2)
    assert s.size == (1920, 1080) and s.n_triangles == 102400
    ref = render_oracle(orc, s, threads=16)
    got = render_hip(s)
    assert (ref["prim"] != 0xFFFFFFFF).sum() > 100000
    assert_frames_equal(got, ref, what="config 2 full size")
    # the same frame lit (the benchmark's configuration list calls this scene "Blinn-Phong + depth";
    # the lit mode is this build's own extension, compared with the oracle's restatement of it)
    ref_lit = render_oracle(orc, s, threads=16, shading=api.SHADE_LIT)
    got_lit = render_hip(s, shading=api.SHADE_LIT)
    assert_frames_equal(got_lit, ref_lit, what="config 2 full size, lit")
    assert np.array_equal(ref_lit["depth"], ref["depth"]) and not np.array_equal(ref_lit["color"], ref["color"])


def test_full_size_config4_single_context_and_eight_bands(orc):
    """Config 4: the 10.24 M-triangle scene at 7680x4320 (32,400 tiles: four chunks of bin offsets,
    33 M pixels), in one context and as the eight sort-first bands of shard.band_rows."""
    from renderer_amd import shard
    s = scenes.config_scene(4)
    w, h = s.size
    assert (w, h) == (7680, 4320)
    ref = orc.render(s, orc.to_model(s.transforms), *orc.camera(s.camera, w, h), w, h, threads=16)
    with api.HipDrawer(w, h, n_images=1, coverage=True) as d:
        d.update_scene(s)
        d.update_camera(s.camera)
        d.draw(0)
        d.sync()
        assert d.stats()["tiles"] == 240 * 135 and d.stats()["prims"] == 10240000
        got = {"color": d.read_color(0), "depth": d.read_depth(0), "prim": d.read_coverage(0)}
    assert_frames_equal(got, ref, what="config 4 single context")
    del got
    covered = np.zeros(h, bool)
    for rank in range(8):
        y0, y1 = shard.band_rows(h, 8, rank)
        with api.HipDrawer(w, h, n_images=1, coverage=True, scissor=(y0, y1)) as d:
            d.update_scene(s)
            d.update_camera(s.camera)
            d.draw(0)
            d.sync()
            band = {"color": d.read_color(0), "depth": d.read_depth(0), "prim": d.read_coverage(0)}
        assert_frames_equal(band, ref, rows=(y0, y1), what="config 4 band %d/8" % rank)
        assert not covered[y0:y1].any()
        covered[y0:y1] = True
    assert covered.all()


def test_full_size_config5_views(orc):
    """Config 5: the 100-sphere scene, 512x512 views from cameras on a sphere around it; ten of the
    4096 views (every rank's first one for eight ranks, and the last two), frames in flight."""
    from renderer_amd import shard
    s = scenes.config_scene(5)
    w, h = s.size
    assert (w, h) == (512, 512) and len(s.objects) == 100
    cams = scenes.orbit_cameras(4096, 5, radius=60.0, target=(0.0, 0.0, -22.5))
    views = [shard.views_for_rank(4096, 8, r)[0] for r in range(8)] + [4094, 4095]
    models = orc.to_model(s.transforms)
    with api.HipDrawer(w, h, n_images=len(views), coverage=True) as d:
        d.update_scene(s)
        for k, v in enumerate(views):
            d.update_camera(cams[v:v + 1])
            d.draw(k)
        d.sync()
        seen = 0
        for k, v in enumerate(views):
            view, proj = orc.camera(cams[v:v + 1], w, h)
            ref = orc.render(s, models, view, proj, w, h)
            got = {"color": d.read_color(k), "depth": d.read_depth(k), "prim": d.read_coverage(k)}
            assert_frames_equal(got, ref, what="config 5 view %d" % v)
            seen += int((ref["prim"] != 0xFFFFFFFF).sum())
    assert seen > 10000


def test_full_size_config5_one_ranks_share_through_view_sets(orc):
    """Config 5 the way a rank runs it: its 512 of the 4096 views (views_for_rank of eight) at
    512x512 through sxe_draw_views, which cuts them into launch sets of 64.  Sixteen of them against
    the oracle, all 512 against update_camera + draw of the same view, bit for bit; the rate of the
    set path is printed (config 5's figure)."""
    import hashlib
    import time
    from renderer_amd import shard
    s = scenes.config_scene(5)
    w, h = s.size
    assert (w, h) == (512, 512) and len(s.objects) == 100
    cams = scenes.orbit_cameras(4096, 5, radius=60.0, target=(0.0, 0.0, -22.5))
    first, end = shard.views_for_rank(4096, 8, 3)
    assert end - first == 512
    cams = np.ascontiguousarray(cams[first:end])
    models = orc.to_model(s.transforms)

    def digest(d, k):
        return hashlib.sha1(d.read_color(k).tobytes() + d.read_depth(k).tobytes()).digest()

    with api.HipDrawer(w, h, n_images=512) as d:
        d.update_scene(s)
        d.draw_views(cams)  # (the first pass sizes the work buffers for sets of 64 views)
        d.sync()
        t0 = time.perf_counter()
        d.draw_views(cams)
        d.sync()
        dt = time.perf_counter() - t0
        print("config 5: 512 views of 512x512 in %.2f ms through sxe_draw_views: %.0f views/s" % (1e3 * dt, 512 / dt))
        by_set = [digest(d, k) for k in range(512)]
        sampled = list(range(0, 512, 37)) + [63, 64, 511]
        for k in sampled[:16]:
            view, proj = orc.camera(cams[k:k + 1], w, h)
            ref = orc.render(s, models, view, proj, w, h)
            assert np.array_equal(d.read_color(k), ref["color"]), "view %d colour" % k
            assert np.array_equal(d.read_depth(k).view(np.uint32), ref["depth"].view(np.uint32)), "view %d depth" % k
        assert d.stats()["redraws"] <= 4
    with api.HipDrawer(w, h, n_images=3) as d:
        d.update_scene(s)
        one_by_one = []
        for k in range(512):
            d.update_camera(cams[k:k + 1])
            d.draw(k % 3)
            d.sync()
            one_by_one.append(digest(d, k % 3))
    differing = [k for k in range(512) if by_set[k] != one_by_one[k]]
    assert not differing, "views %s differ between the set and the single draw" % differing[:8]
    assert len(set(by_set)) > 500  # (they are different views)


def test_config1_lit_cube_at_800x600(orc):
    """Config 1 as BASELINE.json words it ("lit textured cube, 800x600"): the reference's cube, posed
    as the benchmark poses it, textured and lit (the lit mode is this build's extension: parity is
    with the oracle's restatement of the definition in include/sxe.h)."""
    s = scenes.config_scene(1)
    assert s.size == (800, 600) and len(s.objects) == 1
    for shading in (api.SHADE_TEXTURE, api.SHADE_LIT):
        ref = render_oracle(orc, s, shading=shading)
        got = render_hip(s, shading=shading)
        assert_frames_equal(got, ref, what="config 1 at 800x600, shading %d" % shading)
        assert (ref["prim"] != 0xFFFFFFFF).sum() > 50000
    assert tuple(ref["color"][0, 0]) == (218, 124, 0, 255)


def test_views_drawn_as_one_set_equal_views_drawn_one_by_one(orc):
    """sxe_draw_views: the view is a grid dimension of every kernel.  A set of views (more than one
    launch set: 70 views of the small scene) against the oracle and against update_camera + draw,
    with near objects that reach the late-entry and clipping paths in some views."""
    s = scenes.sphere_field_scene(60, 4, 5.0, (-14.0, -4.0), 256, 160, n_textures=2, tex_size=64, name="views")
    w, h = s.size
    n = 70
    cams = scenes.orbit_cameras(n, 9, radius=3.0, target=(0.0, 0.0, -9.0))  # cameras inside the cloud of spheres
    models = orc.to_model(s.transforms)
    with api.HipDrawer(w, h, n_images=n, coverage=True) as d:
        d.update_scene(s)
        d.draw_views(cams)
        d.sync()
        clipped = 0
        for v in range(n):
            view, proj = orc.camera(cams[v:v + 1], w, h)
            ref = orc.render(s, models, view, proj, w, h)
            clipped += ref["stats"]["prims_clipped"]
            got = {"color": d.read_color(v), "depth": d.read_depth(v), "prim": d.read_coverage(v)}
            assert_frames_equal(got, ref, what="view %d of a set" % v)
        assert clipped > 0
        # the same through the single-view entry points, into other images; and a rotation update
        # that is pending when the views are drawn is applied once, before all of them
        last = d.read_color(n - 1)
        d.update_camera(cams[3:4])
        d.draw(0)
        assert np.array_equal(d.read_color(0), d.read_color(3))
        d.rotation_system(3, 0, 400.0)
        d.draw_views(cams[:5], first_image=10)
        t = s.transforms.copy()
        orc.rotation_system(t, 3, 0, 400.0)
        assert d.read_transforms().tobytes() == t.tobytes()
        for v in range(5):
            view, proj = orc.camera(cams[v:v + 1], w, h)
            ref = orc.render(s, orc.to_model(t), view, proj, w, h)
            got = {"color": d.read_color(10 + v), "depth": d.read_depth(10 + v), "prim": d.read_coverage(10 + v)}
            assert_frames_equal(got, ref, what="rotated view %d" % v)
        assert np.array_equal(d.read_color(n - 1), last)
        with pytest.raises(api.SxeError):
            d.draw_views(cams, first_image=1)  # one image short


def test_unrendered_objects_and_two_meshes_over_three_texture_groups(orc):
    """update_drawables' draw list (src/renderer/drawer.rs:313-347,481-487): objects with
    `rendered` false are left out, the rest is drawn by texture group; meshes of different sizes."""
    s = scenes.mixed_scene()
    assert set(np.unique(s.objects["texture"])) == {0, 1, 2} and set(np.unique(s.objects["mesh"])) == {0, 1}
    assert 0 < (s.objects["rendered"] == 0).sum() < len(s.objects)
    ref = render_oracle(orc, s)
    assert ref["stats"]["prims"] == s.n_triangles
    got = render_hip(s, use_drawables=True)
    assert_frames_equal(got, ref, what="mixed scene via update_drawables")
    got = render_hip(s)
    assert_frames_equal(got, ref, what="mixed scene via update_instances")
    # primitive ids follow the draw order: the first drawn object is the first rendered one of texture 0
    everything = scenes.mixed_scene()
    everything.objects["rendered"] = 1
    full = render_oracle(orc, everything)
    assert not np.array_equal(full["color"], ref["color"])


# ---- limits ----

def test_texture_table_that_shrinks_under_a_scene_is_refused(orc):
    s = scenes.reference_demo_scene(width=256, height=192)  # samples textures 0..2
    with api.HipDrawer(256, 192) as d:
        d.update_scene(s)
        d.update_camera(s.camera)
        d.draw(0)
        d.sync()
        before = d.read_color(0)
        d.set_textures(s.textures[:1])  # the scene still refers to textures 1 and 2
        with pytest.raises(api.SxeError) as e:
            d.draw(1)
        assert e.value.code == api.E_STATE and "texture table" in str(e.value)
        with pytest.raises(api.SxeError):
            d.draw_noalloc(1)
        d.set_textures([])
        with pytest.raises(api.SxeError) as e:
            d.draw(1)
        assert e.value.code == api.E_STATE
        d.update_scene(s)  # uploads table and scene together again
        d.draw(1)
        assert np.array_equal(d.read_color(1), before)
        d.set_textures(s.textures + s.textures[:1])  # a larger table keeps the scene valid
        d.draw(0)
        assert np.array_equal(d.read_color(0), before)


def test_more_than_2_to_the_32_bin_entries_is_a_capacity_error():
    """66,000 screen-filling triangles on an 8192x8192 frame touch 65,536 tiles each: 4.33e9
    (triangle, tile) pairs.  The total must not wrap into something that fits."""
    from tests.util import soup_scene
    n = 66000
    tris = np.zeros((n, 3, 3), np.float32)
    tris[:, 0, :2] = (-3.0, -1.0)
    tris[:, 1, :2] = (1.0, -1.0)
    tris[:, 2, :2] = (1.0, 3.0)
    tris[..., 2] = np.linspace(0.1, 0.9, n, dtype=np.float32)[:, None]
    s = soup_scene(tris, width=8192, height=8192)
    ident = np.eye(4, dtype=np.float32).reshape(16)
    with api.HipDrawer(8192, 8192, n_images=1) as d:
        d.update_scene(s)
        d.update_camera_matrices(ident, ident)
        d.draw(0)
        with pytest.raises(api.SxeError) as e:
            d.sync()
        assert e.value.code == api.E_CAPACITY and "2^32" in str(e.value)
        # the context survives: a scene that fits renders
        few = soup_scene(tris[:8], width=8192, height=8192)
        d.update_scene(few)
        d.draw(0)
        d.sync()
        assert (d.read_depth(0) < 1.0).all()


_THREADS_CHILD = r"""
import sys, threading
import numpy as np
sys.path.insert(0, sys.argv[1])
from renderer_amd import api, scenes
from oracle import oracle
oracle.build(); oracle.lib()
jobs = [scenes.config_scene(2, scale=0.15 + 0.05 * k) for k in range(6)]
go = threading.Barrier(len(jobs))
got, errors = [None] * len(jobs), []
def work(k):
    try:
        s = jobs[k]
        w, h = s.size
        go.wait()  # every thread creates its context and draws its first frame at the same moment
        with api.HipDrawer(w, h, coverage=True) as d:
            d.update_scene(s); d.update_camera(s.camera)
            for frame in range(3):
                d.rotation_system(7 + k, frame, 16.0)
                d.draw(0)
            got[k] = {"color": d.read_color(0), "depth": d.read_depth(0), "prim": d.read_coverage(0), "t": d.read_transforms()}
    except Exception as e:  # (reported by the main thread)
        errors.append("thread %d: %r" % (k, e))
threads = [threading.Thread(target=work, args=(k,)) for k in range(len(jobs))]
[t.start() for t in threads]; [t.join() for t in threads]
assert not errors, errors
for k, s in enumerate(jobs):
    w, h = s.size
    ref = oracle.render(s, oracle.to_model(got[k]["t"]), *oracle.camera(s.camera, w, h), w, h, threads=4)
    for plane in ("color", "prim"):
        assert np.array_equal(ref[plane], got[k][plane]), "thread %d: %s differs" % (k, plane)
    assert np.array_equal(ref["depth"].view(np.uint32), got[k]["depth"].view(np.uint32)), "thread %d: depth differs" % k
print("threads-ok")
"""


def test_contexts_created_and_driven_from_different_threads(orc):
    """sxe.h: distinct contexts are independent and may be driven from different threads.  Six
    threads of a fresh process (so that the once-per-device set-up -- lookup tables, the static work
    array's address -- happens under contention) create a context each at the same moment, draw
    three frames of their own scene and must all get the oracle's frame."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _THREADS_CHILD, root], capture_output=True, text=True, timeout=110)
    assert r.returncode == 0 and "threads-ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def test_sort_first_pipeline_on_one_rank(orc):
    """The overlapped draw/gather loop of shard.SortFirstPipeline with a single rank (its band is the
    whole frame, the gather a device copy): events between the library's stream and the gather
    stream must keep every frame intact although nothing waits on the host inside the loop."""
    import torch
    from renderer_amd import shard
    s = scenes.config_scene(3, scale=0.03)
    w, h = s.size
    view, proj = orc.camera(s.camera, w, h)
    t = s.transforms.copy()
    # frames that outgrow the first work buffers are drawn again when their results are taken,
    # after their bands have left: the pipeline says so instead of handing out a stale frame
    from tests.util import soup_scene
    rng = np.random.default_rng(9)
    tris = np.zeros((3000, 3, 3), np.float32)
    tris[..., :2] = rng.uniform(-1, 1, (3000, 3, 2))
    tris[..., 2] = rng.uniform(0.1, 0.9, (3000, 1))
    crowd = soup_scene(tris, width=512, height=512)
    with api.HipDrawer(512, 512, n_images=2) as d:
        d.update_scene(crowd)
        d.update_camera_matrices(IDENT, IDENT)
        pipe = shard.SortFirstPipeline(d, 512, 512, 1, 0)
        pipe.submit(0)
        pipe.submit(1)
        assert d.stats()["redraws"] == 0
        with pytest.raises(shard.StaleBand):
            pipe.finish()
        assert pipe.stale_frames >= 1 and d.stats()["redraws"] == pipe.stale_frames
        pipe = shard.SortFirstPipeline(d, 512, 512, 1, 0)  # (the buffers have grown: the same again is fine)
        pipe.submit(0)
        frame = pipe.finish()
        assert np.array_equal(frame.cpu().numpy(), orc.render(crowd, IDENT[None], IDENT, IDENT, 512, 512)["color"])
    with api.HipDrawer(w, h, n_images=2) as d:
        d.update_scene(s)
        d.update_camera(s.camera)
        d.draw(0)
        d.sync()
        # the buffers have grown to the scene now: nothing is drawn twice from here on
        pipe = shard.SortFirstPipeline(d, h, w, 1, 0)
        n = 7
        for f in range(n):
            d.rotation_system(3, f, 400.0)
            orc.rotation_system(t, 3, f, 400.0)
            pipe.submit(f)
        frame = pipe.finish()
        assert pipe.stale_frames == 0
        assert pipe.last_gather_ms(n - 1) >= 0.0
        ref = orc.render(s, orc.to_model(t), view, proj, w, h)
        assert np.array_equal(frame.cpu().numpy(), ref["color"])
        assert np.array_equal(d.read_color((n - 1) & 1), ref["color"])


# ---- BlendState::ALPHA with translucent textures (src/renderer/drawer.rs:660-663) ----

def test_translucent_textures_are_blended_in_submission_order(orc):
    """Textures with alpha: the tiles their triangles touch depth-test and blend fragment by fragment
    in submission order; tiles with opaque triangles only keep the fast path.  Cube, sphere field
    with two opaque and two translucent textures, a clip-space soup with clipped fans."""
    c = scenes.cube_scene(320, 240)
    c.textures = [scenes.translucent_texture(128, 3)]
    ref = render_oracle(orc, c)
    got = render_hip(c)
    assert_frames_equal(got, ref, what="translucent cube")
    assert (ref["color"][..., 3] != 255).sum() == 0  # over an opaque clear colour alpha stays 255 ...
    opaque = scenes.cube_scene(320, 240)
    assert not np.array_equal(render_oracle(orc, opaque)["color"], ref["color"])  # ... but the colours show through

    s = scenes.config_scene(2, scale=0.4)
    s.textures = [s.textures[0], scenes.translucent_texture(256, 5), s.textures[2], scenes.translucent_texture(64, 6)]
    ref = render_oracle(orc, s, threads=8)
    got = render_hip(s)
    assert_frames_equal(got, ref, what="spheres, two of four textures translucent")
    # the build of the kernels that keeps no primitive-id image
    with api.HipDrawer(*s.size) as d:
        d.update_scene(s)
        d.update_camera(s.camera)
        d.draw(0)
        assert np.array_equal(d.read_color(0), ref["color"]) and np.array_equal(d.read_depth(0).view(np.uint32), ref["depth"].view(np.uint32))
    for shading in (1, 2):
        ref = render_oracle(orc, s, shading=shading)
        got = render_hip(s, shading=shading)
        assert_frames_equal(got, ref, what="translucent, shading mode %d" % shading)

    from tests.util import random_soup, soup_scene
    rng = np.random.default_rng(31)
    w, h = 200, 120
    soup = soup_scene(random_soup(rng, 900, w, h), width=w, height=h, objects_split=[300, 600], tex_ids=[0, 1],
                      textures=[scenes.checker_texture(64, 5), scenes.translucent_texture(64, 8)])
    ident = np.eye(4, dtype=np.float32).reshape(16)
    models = np.tile(ident, (2, 1))
    ref = orc.render(soup, models, ident, ident, w, h)
    assert ref["stats"]["prims_clipped"] > 30
    got = render_hip(soup, models=models, view=ident, proj=ident)
    assert_frames_equal(got, ref, what="translucent soup")
    # clear colour with alpha below one: the target's alpha channel takes part
    clear = (0.1, 0.3, 0.5, 0.25)
    with api.HipDrawer(320, 240, coverage=True, clear=clear) as d:
        d.update_scene(c)
        d.update_camera(c.camera)
        d.draw(0)
        got = {"color": d.read_color(0), "depth": d.read_depth(0), "prim": d.read_coverage(0)}
    ref = render_oracle(orc, c, clear=clear)
    assert_frames_equal(got, ref, what="translucent cube over a translucent clear colour")
    assert len(np.unique(ref["color"][..., 3])) > 3 and ref["color"][0, 0, 3] == 64


def test_translucent_views_and_tiles_beyond_one_sort_of_the_ordered_path(orc):
    from tests.util import soup_scene
    # a set of views with a translucent texture
    s = scenes.sphere_field_scene(20, 4, 5.0, (-14.0, -4.0), 160, 96, n_textures=2, tex_size=64, name="views")
    s.textures = [s.textures[0], scenes.translucent_texture(64, 9)]
    cams = scenes.orbit_cameras(5, 9, radius=12.0, target=(0.0, 0.0, -9.0))
    models = orc.to_model(s.transforms)
    with api.HipDrawer(160, 96, n_images=5, coverage=True) as d:
        d.update_scene(s)
        d.draw_views(cams)
        for v in range(5):
            view, proj = orc.camera(cams[v:v + 1], 160, 96)
            ref = orc.render(s, models, view, proj, 160, 96)
            got = {"color": d.read_color(v), "depth": d.read_depth(v), "prim": d.read_coverage(v)}
            assert_frames_equal(got, ref, what="translucent view %d" % v)
    # tiles of one to four thousand entries (every sorting thread then holds several keys)
    ident = np.eye(4, dtype=np.float32).reshape(16)
    for n, seed in ((9000, 5), (19000, 6)):
        rng = np.random.default_rng(seed)
        tris = np.zeros((n, 3, 3), np.float32)
        tris[..., :2] = rng.uniform(-0.2, 0.2, (n, 1, 2)) + rng.uniform(-0.03, 0.03, (n, 3, 2))
        tris[..., 2] = rng.uniform(0.1, 0.9, (n, 1))
        crowd = soup_scene(tris, width=128, height=128, textures=[scenes.translucent_texture(64, 2)])
        ref = orc.render(crowd, ident[None], ident, ident, 128, 128)
        with api.HipDrawer(128, 128, coverage=True) as d:
            d.update_scene(crowd)
            d.update_camera_matrices(ident, ident)
            d.draw(0)
            got = {"color": d.read_color(0), "depth": d.read_depth(0), "prim": d.read_coverage(0)}
            assert 1024 < d.tile_counts().max() <= 4096
        assert_frames_equal(got, ref, what="crowded translucent tiles, %d triangles" % n)
    # more than 4096 entries in a tile that needs ordering: sorted and blended in consecutive
    # stretches of the submission order (BlendState::ALPHA knows no such limit, drawer.rs:660-663);
    # with clipped fans (several entries of one rank) and a second, larger crowd
    for n, seed, spread in ((32000, 77, 0.2), (60000, 78, 0.12)):
        rng = np.random.default_rng(seed)
        tris = np.zeros((n, 3, 3), np.float32)
        tris[..., :2] = rng.uniform(-spread, spread, (n, 1, 2)) + rng.uniform(-0.03, 0.03, (n, 3, 2))
        tris[..., 2] = rng.uniform(0.1, 0.9, (n, 1))
        tris[::97, :, 2] = rng.uniform(-0.4, 1.4, (len(tris[::97]), 3))  # some cross the near or the far plane
        dense = soup_scene(tris, width=128, height=128, textures=[scenes.translucent_texture(64, 2)])
        ref = orc.render(dense, ident[None], ident, ident, 128, 128)
        with api.HipDrawer(128, 128, coverage=True) as d:
            d.update_scene(dense)
            d.update_camera_matrices(ident, ident)
            d.draw(0)
            d.sync()
            got = {"color": d.read_color(0), "depth": d.read_depth(0), "prim": d.read_coverage(0)}
            assert d.tile_counts().max() > 4096 * (1 if n == 32000 else 2)
        assert_frames_equal(got, ref, what="translucent tiles beyond one sort, %d triangles" % n)


def test_ui_quad_layer(orc):
    """The UI pipeline (ui.vert:11-15, ui.frag:12) on the widget of src/primitives/two_d/widget.rs:16-26:
    positions passed through (identity matrices), vertex colour times texture; the unrendered main
    box is skipped, overlapping quads at depth 0 keep the first one drawn."""
    root = scenes.esc_menu()
    s = scenes.ui_scene(root, [scenes.checker_texture(64, 4), scenes.translucent_texture(64, 7)], texture_of=[0, 0, 1],
                        width=640, height=480)
    ident = np.eye(4, dtype=np.float32).reshape(16)
    models = np.tile(ident, (3, 1))
    for tex_of in ([0, 0, 0], [0, 0, 1]):  # opaque; the second child translucent (ordered path)
        s.objects["texture"] = tex_of
        ref = orc.render(s, models, ident, ident, 640, 480, shading=2)
        got = render_hip(s, view=ident, proj=ident, shading=2, use_drawables=True)
        assert (ref["prim"] != 0xFFFFFFFF).sum() > 30000
        assert_frames_equal(got, ref, what="ui quads, textures %s" % tex_of)
