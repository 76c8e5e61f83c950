"""A context per target image (api.FramesInFlight): the reference keeps a command buffer and a fence
per swapchain image and submits the next frame while the last is still being rendered
(src/renderer/drawer.rs:412-456).  The frames of different images overlap on the GPU here; each must
still be the frame the oracle draws for its place in the animation."""
import numpy as np
import pytest

from renderer_amd import api, scenes
from tests.util import assert_frames_equal


@pytest.mark.gpu
@pytest.mark.parametrize("n_images", [2, 3])
def test_overlapping_frames_are_the_frames_of_one_context(orc, n_images):
    s = scenes.config_scene(3, scale=0.03)
    w, h = s.size
    view, proj = orc.camera(s.camera, w, h)
    t = s.transforms.copy()
    with api.FramesInFlight(w, h, n_images=n_images, coverage=True) as d:
        d.update_scene(s)
        d.update_camera(s.camera)
        refs = {}
        for f in range(9):
            d.rotation_system(3, f, 400.0)  # large steps: consecutive frames differ visibly
            orc.rotation_system(t, 3, f, 400.0)
            d.draw(f % n_images)  # no host wait between the frames
            refs[f % n_images] = orc.render(s, orc.to_model(t), view, proj, w, h)
        d.sync()
        for img, want in refs.items():
            got = {"color": d.read_color(img), "depth": d.read_depth(img), "prim": d.read_coverage(img)}
            assert_frames_equal(got, want, what="image %d of %d" % (img, n_images))
        # every context has applied, or still holds, the same updates: the state read back is the oracle's
        assert np.array_equal(d.read_transforms().view(np.uint8), t.view(np.uint8))

        # two updates without a frame between them, then new transforms for everything (the updates
        # still waiting in the other contexts are void), then the animation goes on
        d.rotation_system(3, 9, 400.0)
        d.rotation_system(4, 0, 250.0)
        t = s.transforms.copy()
        t["position"][:, 0] += 0.5
        d.update_transforms(t)
        for f in range(10, 10 + n_images + 1):
            d.rotation_system(3, f, 400.0)
            orc.rotation_system(t, 3, f, 400.0)
            d.draw(f % n_images)
            refs[f % n_images] = orc.render(s, orc.to_model(t), view, proj, w, h)
        d.sync()
        for img, want in refs.items():
            got = {"color": d.read_color(img), "depth": d.read_depth(img), "prim": d.read_coverage(img)}
            assert_frames_equal(got, want, what="after new transforms, image %d" % img)


@pytest.mark.gpu
def test_overlapping_frames_with_updates_that_are_not_consecutive(orc):
    """Updates a context has missed are applied as spans where they are consecutive frames of one
    seed and step, and one by one where they are not."""
    s = scenes.config_scene(2, scale=0.2)
    w, h = s.size
    view, proj = orc.camera(s.camera, w, h)
    t = s.transforms.copy()
    with api.FramesInFlight(w, h, n_images=2, coverage=True) as d:
        d.update_scene(s)
        d.update_camera(s.camera)
        updates = [(3, 0, 300.0), (3, 1, 300.0), (3, 5, 300.0), (8, 6, 300.0), (8, 7, 120.0), (8, 8, 120.0)]
        for k, (seed, frame, delta) in enumerate(updates):
            d.rotation_system(seed, frame, delta)
            orc.rotation_system(t, seed, frame, delta)
            if k in (1, 4, 5):
                d.draw(k & 1)
                want = orc.render(s, orc.to_model(t), view, proj, w, h)
                d.sync()
                got = {"color": d.read_color(k & 1), "depth": d.read_depth(k & 1), "prim": d.read_coverage(k & 1)}
                assert_frames_equal(got, want, what="after update %d" % k)
    with pytest.raises(api.SxeError):
        api.FramesInFlight(w, h, n_images=2).draw(2)


@pytest.mark.gpu
def test_overlapping_frames_at_full_size(orc):
    """Config 3 as the bench draws it (10.24 M triangles, 3840x2160): frames of 0.2 ms that do overlap
    on the GPU.  The last frame of each image equals the frame a single context draws for that place
    in the animation, and the very last one the oracle's."""
    s = scenes.config_scene(3)
    w, h = s.size
    frames = 8
    with api.FramesInFlight(w, h, n_images=2) as d:
        d.update_scene(s)
        d.update_camera(s.camera)
        for f in range(frames):
            d.rotation_system(3, f, 16.0)
            d.draw(f & 1)
        d.sync()
        got = {img: (d.read_color(img), d.read_depth(img)) for img in (0, 1)}
    with api.HipDrawer(w, h, n_images=2) as one:
        one.update_scene(s)
        one.update_camera(s.camera)
        for f in range(frames):
            one.rotation_system(3, f, 16.0)
            one.draw(f & 1)
            if f >= frames - 2:
                one.sync()
                color, depth = one.read_color(f & 1), one.read_depth(f & 1)
                assert np.array_equal(color, got[f & 1][0]), "colour of frame %d" % f
                assert np.array_equal(depth.view(np.uint32), got[f & 1][1].view(np.uint32)), "depth of frame %d" % f
        t = one.read_transforms()
    view, proj = orc.camera(s.camera, w, h)
    want = orc.render(s, orc.to_model(t), view, proj, w, h, threads=16)
    last = (frames - 1) & 1
    assert np.array_equal(want["color"], got[last][0])
    assert np.array_equal(want["depth"].view(np.uint32), got[last][1].view(np.uint32))
