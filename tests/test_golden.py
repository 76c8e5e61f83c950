"""Golden fixtures (tests/golden/*.npz, written by tests/golden/make_golden.py from the oracle):
the oracle must keep reproducing them bit for bit (CPU), and so must the HIP library (GPU)."""
import glob
import os

import numpy as np
import pytest

from renderer_amd import api, scenes

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = sorted(glob.glob(os.path.join(HERE, "golden", "*.npz")))


def load(path):
    z = np.load(path, allow_pickle=False)
    texs = []
    while "texture%d" % len(texs) in z:
        texs.append(z["texture%d" % len(texs)])
    w, h = (int(x) for x in z["size"])
    s = scenes.Scene(z["vertices"], z["indices"], z["meshes"], z["objects"], z["transforms"], texs,
                     camera=z["camera"], size=(w, h), name=os.path.basename(path))
    return s, z


def test_fixtures_exist():
    assert len(FIXTURES) >= 4


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p) for p in FIXTURES])
def test_oracle_reproduces_golden(orc, path):
    s, z = load(path)
    w, h = s.size
    # (a fixture rendered with the draw list of the reference binary says so)
    r = orc.render(s, z["models"], z["view"], z["proj"], w, h, flags=orc.FLAG_REFERENCE_INDEXING if "reference_indexing" in z else 0)
    assert np.array_equal(r["prim"], z["prim"])
    assert np.array_equal(r["depth"].view(np.uint32), z["depth_bits"])
    assert np.array_equal(r["color"], z["color"])
    if "soup" not in path:
        # the stored matrices are what the oracle's Transform / camera maths produce
        assert orc.to_model(s.transforms).tobytes() == z["models"].tobytes()
        v, p = orc.camera(s.camera, w, h)
        assert v.tobytes() == z["view"].tobytes() and p.tobytes() == z["proj"].tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p) for p in FIXTURES])
def test_hip_reproduces_golden(path):
    s, z = load(path)
    w, h = s.size
    with api.HipDrawer(w, h, coverage=True, reference_indexing="reference_indexing" in z) as d:
        d.update_scene(s)
        if "soup" in path:
            d.update_uniforms(z["models"])
            d.update_camera_matrices(z["view"], z["proj"])
        else:
            d.update_camera(s.camera)  # transforms -> models and camera -> view/proj inside the library
        d.draw(0)
        d.sync()
        assert np.array_equal(d.read_coverage(0), z["prim"])
        assert np.array_equal(d.read_depth(0).view(np.uint32), z["depth_bits"])
        assert np.array_equal(d.read_color(0), z["color"])
        if "soup" not in path:
            assert d.read_models().tobytes() == z["models"].tobytes()
