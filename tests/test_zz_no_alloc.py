"""sxe_draw_noalloc (every work buffer in the library's static device array) against sxe_draw,
through the C ABI: two calls in a row on one GPU, the second with a smaller scene, so that the
second call meets the first one's leftovers in the shared array."""
import numpy as np
import pytest

from renderer_amd import api, scenes

pytestmark = pytest.mark.gpu


def _images(d, img):
    return {"color": d.read_color(img), "depth": d.read_depth(img), "prim": d.read_coverage(img)}


def _same(a, b, what):
    assert np.array_equal(a["prim"], b["prim"]), what + ": coverage differs"
    assert np.array_equal(a["depth"].view(np.uint32), b["depth"].view(np.uint32)), what + ": depth bits differ"
    assert np.array_equal(a["color"], b["color"]), what + ": colour differs"


def test_noalloc_draw_equals_draw_twice_in_a_row_second_smaller():
    from tests.util import random_soup, soup_scene
    big = scenes.config_scene(2, scale=0.5)      # 50 spheres, 960 x 540
    small = scenes.config_scene(2, scale=0.2)    # 20 spheres, 384 x 216
    rng = np.random.default_rng(12)
    soup = soup_scene(random_soup(rng, 900, 160, 96), width=160, height=96)  # clipped fans, huge and tiny triangles
    ident = np.eye(4, dtype=np.float32).reshape(16)
    ctxs = []
    try:
        for s in (big, small, soup):
            d = api.HipDrawer(s.size[0], s.size[1], n_images=2, coverage=True)
            d.update_scene(s)
            if s is soup:
                d.update_camera_matrices(ident, ident)
            else:
                d.update_camera(s.camera)
            d.draw(0)
            d.sync()
            ctxs.append((d, _images(d, 0)))
        # everything is allocated and uploaded: now the calls under test, back to back
        redraws = [d.stats()["redraws"] for d, _ in ctxs]
        for d, _ in ctxs:
            d.draw_noalloc(1)
        for (d, ref), name in zip(ctxs, ("big", "small", "soup")):
            assert (ref["prim"] != 0xFFFFFFFF).sum() > 500
            _same(_images(d, 1), ref, name)
        assert [d.stats()["redraws"] for d, _ in ctxs] == redraws  # nothing grew, nothing was drawn again
        # and once more in the other order, with a rotation update riding on the frame
        for d, _ in reversed(ctxs[:2]):
            d.rotation_system(3, 0, 400.0)
            d.draw_noalloc(1)
        for (d, ref), name in zip(ctxs[:2], ("big", "small")):
            got = _images(d, 1)
            assert not np.array_equal(got["color"], ref["color"])
            d.draw(0)  # same (already rotated) transforms through the allocating entry point
            d.sync()
            _same(got, _images(d, 0), name + " rotated")
    finally:
        for d, _ in ctxs:
            d.close()


def test_noalloc_rejects_what_does_not_fit_and_frames_in_flight():
    s = scenes.config_scene(3, scale=0.05)  # 500 spheres = 512,000 triangles: beyond the static array
    with api.HipDrawer(s.size[0], s.size[1]) as d:
        d.update_scene(s)
        d.update_camera(s.camera)
        with pytest.raises(api.SxeError) as e:
            d.draw_noalloc(0)
        assert e.value.code == api.E_CAPACITY and "static work array" in str(e.value)
        d.draw(0)  # the allocating entry point still takes the scene
        d.sync()
        assert (d.read_depth(0) < 1.0).sum() > 100
    c = scenes.cube_scene(160, 120)
    with api.HipDrawer(160, 120) as d:
        with pytest.raises(api.SxeError) as e:
            d.draw_noalloc(0)
        assert e.value.code == api.E_STATE
        d.update_scene(c)
        d.update_camera(c.camera)
        d.draw(0)
        with pytest.raises(api.SxeError) as e:
            d.draw_noalloc(1)  # a frame of sxe_draw is in flight
        assert e.value.code == api.E_STATE
        d.sync()
        d.draw_noalloc(1)
        assert np.array_equal(d.read_color(0), d.read_color(1))
