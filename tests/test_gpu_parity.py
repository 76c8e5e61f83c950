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
