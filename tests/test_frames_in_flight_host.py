"""Host logic of api.FramesInFlight without a GPU: which context gets which call, and how the
rotation updates a context has missed reach it (runs of consecutive frames as one span)."""
import pytest

from renderer_amd import api


class _Recorder:
    """Stands in for api.HipDrawer: records the calls it receives."""
    made = []

    def __init__(self, width, height, n_images=2, **kw):
        self.calls, self.kw, self.n_images = [], kw, n_images
        _Recorder.made.append(self)

    def __getattr__(self, name):
        def call(*a, **kw):
            self.calls.append((name,) + a + tuple(sorted(kw.items())))
            return {"redraws": 1} if name == "stats" else None
        return call


@pytest.fixture
def pair(monkeypatch):
    _Recorder.made = []
    monkeypatch.setattr(api, "HipDrawer", _Recorder)
    d = api.FramesInFlight(64, 48, n_images=2, coverage=True)
    assert [c.n_images for c in _Recorder.made] == [1, 1] and _Recorder.made[0].kw == {"coverage": True}
    return d, _Recorder.made


def test_state_goes_to_every_context_and_frames_to_their_own(pair):
    d, (a, b) = pair
    d.update_camera("cam")
    d.draw(0)
    d.draw(1)
    d.draw(1)
    assert a.calls == [("update_camera", "cam"), ("draw", 0)]
    assert b.calls == [("update_camera", "cam"), ("draw", 0), ("draw", 0)]  # (every context draws into its only image)
    for bad in (2, -1):
        with pytest.raises(api.SxeError):
            d.draw(bad)
    d.read_color(1)
    d.read_depth(0)
    assert b.calls[-1] == ("read_color", 0) and a.calls[-1] == ("read_depth", 0)
    assert d.stats()["redraws"] == 2  # (summed over the contexts)


def test_missed_updates_arrive_as_spans(pair):
    d, (a, b) = pair
    for f in range(4):
        d.rotation_system(3, f, 16.0)
        d.draw(f & 1)
    # context a drew frames 0 and 2, b frames 1 and 3: each caught up on what it had not seen
    assert a.calls == [("rotation_system", 3, 0, 16.0, ("n_frames", 1)), ("draw", 0),
                       ("rotation_system", 3, 1, 16.0, ("n_frames", 2)), ("draw", 0)]
    assert b.calls == [("rotation_system", 3, 0, 16.0, ("n_frames", 2)), ("draw", 0),
                       ("rotation_system", 3, 2, 16.0, ("n_frames", 2)), ("draw", 0)]
    # a gap in the frame numbers, another seed, another step: separate launches, in order
    a.calls.clear()
    for seed, frame, delta in [(3, 4, 16.0), (3, 5, 16.0), (3, 9, 16.0), (5, 10, 16.0), (5, 11, 8.0), (5, 12, 8.0)]:
        d.rotation_system(seed, frame, delta)
    d.rotation_system(5, 13, 8.0, n_frames=3)  # (a span joins the run it continues)
    d.draw(0)
    # (frame 3, which the other context drew, was still waiting for this one)
    assert a.calls == [("rotation_system", 3, 3, 16.0, ("n_frames", 3)), ("rotation_system", 3, 9, 16.0, ("n_frames", 1)),
                       ("rotation_system", 5, 10, 16.0, ("n_frames", 1)), ("rotation_system", 5, 11, 8.0, ("n_frames", 5)),
                       ("draw", 0)]


def test_new_transforms_void_the_updates_still_waiting(pair):
    d, (a, b) = pair
    d.rotation_system(3, 0, 16.0)
    d.draw(0)                      # a has applied frame 0, b has not
    d.rotation_system(3, 1, 16.0)
    d.update_transforms("t")       # replaces every transform: what was waiting has nothing left to act on
    d.rotation_system(3, 2, 16.0)
    d.draw(1)
    assert b.calls == [("update_transforms", "t"), ("rotation_system", 3, 2, 16.0, ("n_frames", 1)), ("draw", 0)]
    d.read_transforms()            # the first context catches up before it is read
    assert a.calls[-2:] == [("rotation_system", 3, 2, 16.0, ("n_frames", 1)), ("read_transforms",)]


def test_a_context_that_cannot_be_created_closes_the_others(monkeypatch):
    made = []

    class Failing(_Recorder):
        def __init__(self, *a, **kw):
            if len(made) == 1:
                raise api.SxeError(api.E_DEVICE, "no device")
            super().__init__(*a, **kw)
            made.append(self)

    monkeypatch.setattr(api, "HipDrawer", Failing)
    with pytest.raises(api.SxeError):
        api.FramesInFlight(64, 48, n_images=2)
    assert made[0].calls == [("close",)]
    with pytest.raises(api.SxeError):
        api.FramesInFlight(64, 48, n_images=0)
