"""Byte order of the colour target.  The reference renders into the first sRGB format its surface
lists (src/renderer/presenter.rs:472-478): Bgra8Srgb on X11, else its fallback Rgba8Srgb.  The values
are the same, red and blue change places in memory (SXE_FLAG_BGRA / ORC_FLAG_BGRA)."""
import numpy as np
import pytest

from renderer_amd import api, scenes


def _scenes():
    opaque = scenes.config_scene(2, scale=0.15)
    blended = scenes.cube_scene(200, 150)
    blended.textures = [scenes.translucent_texture(64, 3)]
    return [opaque, blended]


def _orc(orc, s, flags):
    w, h = s.size
    view, proj = orc.camera(s.camera, w, h)
    return orc.render(s, orc.to_model(s.transforms), view, proj, w, h, flags=flags)


def test_oracle_bgra_is_rgba_with_red_and_blue_exchanged(orc):
    for s in _scenes():
        rgba, bgra = _orc(orc, s, 0), _orc(orc, s, orc.FLAG_BGRA)
        assert np.array_equal(bgra["color"], rgba["color"][..., [2, 1, 0, 3]])
        assert np.array_equal(bgra["depth"], rgba["depth"]) and np.array_equal(bgra["prim"], rgba["prim"])
        assert (rgba["color"][..., 0] != rgba["color"][..., 2]).any()
    # the clear colour (0.7, 0.2, 0.0, 1.0) -> (218, 124, 0, 255) comes out as (0, 124, 218, 255)
    assert tuple(_orc(orc, _scenes()[0], orc.FLAG_BGRA)["color"][0, 0]) == (0, 124, 218, 255)


@pytest.mark.gpu
def test_hip_bgra_equals_oracle(orc):
    for s in _scenes():
        w, h = s.size
        want = _orc(orc, s, orc.FLAG_BGRA)
        with api.HipDrawer(w, h, coverage=True, bgra=True) as d:
            d.update_scene(s)
            d.update_camera(s.camera)
            d.draw(0)
            d.sync()
            assert np.array_equal(d.read_color(0), want["color"]), s.name
            assert np.array_equal(d.read_depth(0).view(np.uint32), want["depth"].view(np.uint32))
            assert np.array_equal(d.map_color(0), want["color"])


@pytest.mark.gpu
def test_depth_images_that_are_not_kept(orc):
    """SXE_FLAG_DISCARD_DEPTH, the reference's AttachmentStoreOp::DontCare for the depth attachment
    (src/renderer/drawer.rs:544-547): colour and coverage are what they are with the depth images
    kept (the test inside the frame is the same), on the opaque and on the ordered path and for the
    allocation-free draw; reading the depth back is refused."""
    for s in _scenes():
        w, h = s.size
        want = _orc(orc, s, 0)
        with api.HipDrawer(w, h, coverage=True, discard_depth=True) as d:
            d.update_scene(s)
            d.update_camera(s.camera)
            d.draw(0)
            d.sync()
            assert np.array_equal(d.read_color(0), want["color"]), s.name
            assert np.array_equal(d.read_coverage(0), want["prim"])
            with pytest.raises(api.SxeError) as e:
                d.read_depth(0)
            assert e.value.code == api.E_STATE
            d.draw_noalloc(1)
            assert np.array_equal(d.read_color(1), want["color"])
