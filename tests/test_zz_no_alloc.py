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


def test_noalloc_calls_that_overlap_on_one_gpu_are_refused_not_raced():
    """Two threads, two contexts, one GPU, one static array: a call that finds the other one in
    progress returns SXE_E_STATE and draws nothing; every call that went through gives the frame
    sxe_draw gives."""
    import threading
    scenes_ = [scenes.config_scene(2, scale=0.5), scenes.config_scene(2, scale=0.3)]
    ctxs = []
    try:
        for s in scenes_:
            d = api.HipDrawer(s.size[0], s.size[1], n_images=2, coverage=True)
            d.update_scene(s)
            d.update_camera(s.camera)
            d.draw(0)
            d.sync()
            ctxs.append((d, _images(d, 0)))
        outcome = [{"ok": 0, "busy": 0, "bad": []} for _ in ctxs]
        start = threading.Barrier(len(ctxs))

        def worker(k):
            d, ref = ctxs[k]
            start.wait()
            for _ in range(40):
                try:
                    d.draw_noalloc(1)  # (ctypes releases the interpreter lock for the duration of the call)
                except api.SxeError as e:
                    if e.code == api.E_STATE and "another sxe_draw_noalloc" in str(e):
                        outcome[k]["busy"] += 1
                    else:
                        outcome[k]["bad"].append(str(e))
                    continue
                outcome[k]["ok"] += 1
                got = _images(d, 1)
                if not (np.array_equal(got["color"], ref["color"]) and np.array_equal(got["depth"], ref["depth"])):
                    outcome[k]["bad"].append("frame differs")
        threads = [threading.Thread(target=worker, args=(k,)) for k in range(len(ctxs))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        for o in outcome:
            assert not o["bad"], o["bad"][:3]
            assert o["ok"] + o["busy"] == 40
        # (one thread may lose every race; some call went through, and with the other thread gone
        # every context's call does)
        assert sum(o["ok"] for o in outcome) >= 1
        for d, ref in ctxs:
            d.draw_noalloc(1)
            got = _images(d, 1)
            assert np.array_equal(got["color"], ref["color"]) and np.array_equal(got["depth"], ref["depth"])
    finally:
        for d, _ in ctxs:
            d.close()


def test_noalloc_says_when_the_tile_count_is_the_limit_and_tile_diagnostics_follow_the_frame():
    c = scenes.cube_scene(7680, 4320)  # 32,400 tiles: no room for bins, however small the scene
    with api.HipDrawer(7680, 4320, n_images=1) as d:
        d.update_scene(c)
        d.update_camera(c.camera)
        with pytest.raises(api.SxeError) as e:
            d.draw_noalloc(0)
        assert e.value.code == api.E_CAPACITY and "screen tiles" in str(e.value)
    s = scenes.config_scene(2, scale=0.3)
    with api.HipDrawer(s.size[0], s.size[1]) as d:
        d.update_scene(s)
        d.update_camera(s.camera)
        d.draw(0)
        d.sync()
        want = d.tile_counts().copy()
        small = scenes.cube_scene(s.size[0], s.size[1])
        d.update_scene(small)
        d.update_camera(small.camera)
        d.draw_noalloc(0)
        got = d.tile_counts()  # of the frame just drawn, which worked in the static array
        assert got.sum() > 0 and got.sum() < want.sum()
        d.draw(0)
        d.sync()
        assert np.array_equal(d.tile_counts(), got)
