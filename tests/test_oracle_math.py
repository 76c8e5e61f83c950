"""Pins the oracle's restatement of the reference's matrix maths with closed-form known answers.

The reference has no tests; these constants are derived from the formulas it calls
(src/components/transform.rs:23-80, src/renderer/drawer.rs:242-258,390-409 and cgmath 0.17.0's
documented definitions) and were computed independently in float64 below.
"""
import math

import numpy as np
import pytest


def test_trig_matches_libm(orc):
    xs = np.concatenate([np.linspace(-20.0, 20.0, 4001), np.radians(np.arange(-720, 721, 0.25)) * 0.5,
                         [0.0, 1e-8, -1e-8, math.pi, math.pi / 2, 3 * math.pi / 4, 1000.5, -999.25]])
    for x in xs.astype(np.float32):
        s, c = orc.sincos(float(x))
        assert abs(s - math.sin(float(x))) < 3e-7 and abs(c - math.cos(float(x))) < 3e-7, x
    for y, x in [(0, 0), (0, 1), (0, -1), (1, 0), (-1, 0), (1, 1), (-1, 1), (1, -1), (-1, -1), (0.3, -2.0),
                 (5.0, 0.001), (-1e-3, -7.0), (2.5, 1.0), (0.41, 1.0), (0.42, 1.0), (2.41, 1.0), (2.42, 1.0)]:
        assert abs(orc.atan2(y, x) - math.atan2(y, x)) < 4e-7, (y, x)
    for x in np.linspace(-1, 1, 801).astype(np.float32):
        assert abs(orc.asin(float(x)) - math.asin(float(x))) < 5e-7, x
    assert orc.asin(2.0) == pytest.approx(math.pi / 2) and orc.asin(-2.0) == pytest.approx(-math.pi / 2)


def _quat_from_euler_deg64(x, y, z):
    sx, cx = math.sin(math.radians(x) / 2), math.cos(math.radians(x) / 2)
    sy, cy = math.sin(math.radians(y) / 2), math.cos(math.radians(y) / 2)
    sz, cz = math.sin(math.radians(z) / 2), math.cos(math.radians(z) / 2)
    return np.array([-sx * sy * sz + cx * cy * cz, sx * cy * cz + sy * sz * cx,
                     -sx * sz * cy + sy * cx * cz, sx * sy * cz + sz * cx * cy])


def _rot64(q):
    s, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - s * z), 2 * (x * z + s * y)],
                     [2 * (x * y + s * z), 1 - 2 * (x * x + z * z), 2 * (y * z - s * x)],
                     [2 * (x * z - s * y), 2 * (y * z + s * x), 1 - 2 * (x * x + y * y)]])


def test_quaternion_from_euler_and_product(orc):
    rng = np.random.default_rng(0)
    for _ in range(200):
        e = rng.uniform(-360, 360, 3)
        q = orc.quat_from_euler_deg(*e)
        assert np.allclose(q, _quat_from_euler_deg64(*e), atol=3e-7)
        assert abs(np.linalg.norm(q) - 1) < 1e-6
    # identity transform: Quaternion::from(Euler 0,0,0) = (1,0,0,0) (transform.rs:28)
    t = orc.identity_transforms(1)
    assert t["rotation"].tolist() == [[1.0, 0.0, 0.0, 0.0]]
    # Hamilton product, right-multiplied (transform.rs:44): i*j = k
    assert orc.quat_mul([0, 1, 0, 0], [0, 0, 1, 0]).tolist() == [0, 0, 0, 1]
    assert orc.quat_mul([0, 0, 1, 0], [0, 1, 0, 0]).tolist() == [0, 0, 0, -1]
    # rotate() composes on the right and does not renormalise
    orc.rotate(t, 0, 90.0, 0.0, 0.0)
    orc.rotate(t, 0, 0.0, 90.0, 0.0)
    want = orc.quat_mul(orc.quat_from_euler_deg(90, 0, 0), orc.quat_from_euler_deg(0, 90, 0))
    assert np.array_equal(t["rotation"][0], want)


def test_model_matrix_is_T_R_S(orc):
    rng = np.random.default_rng(1)
    t = orc.identity_transforms(50)
    t["position"] = rng.uniform(-20, 20, (50, 3))
    t["scale"] = rng.uniform(0.2, 3, (50, 3))
    t = np.ascontiguousarray(t)
    for i in range(50):
        orc.rotate(t, i, *rng.uniform(-180, 180, 3))
    m = orc.to_model(t).reshape(50, 4, 4)  # [col][row]
    for i in range(50):
        R = _rot64(t["rotation"][i].astype(np.float64))
        want = np.eye(4)
        want[:3, :3] = R * t["scale"][i][None, :]
        want[:3, 3] = t["position"][i]
        assert np.allclose(m[i].T, want, atol=2e-5)
        assert m[i][0][3] == 0 and m[i][3][3] == 1  # bottom row (0,0,0,1)


def test_projection_known_answers(orc):
    # closed-form constants: cot(22.5 deg) = 2.41421356...
    p43 = orc.perspective(45.0, 800.0 / 600.0, 0.1, 1000.0).reshape(4, 4)
    assert p43[0][0] == pytest.approx(1.81066, abs=2e-6)
    assert p43[1][1] == pytest.approx(-2.41421342, abs=5e-7)  # Y flip only (drawer.rs:255)
    assert p43[2][2] == pytest.approx(-1.00019991, abs=2e-7)   # OpenGL z range kept
    assert p43[2][3] == -1.0
    assert p43[3][2] == pytest.approx(-0.20002, abs=2e-7)
    assert p43[3][3] == 0.0
    assert orc.perspective(45.0, 16.0 / 9.0, 0.1, 1000.0)[0] == pytest.approx(1.35799503, abs=5e-7)
    assert orc.perspective(45.0, 1.0, 0.1, 1000.0)[0] == pytest.approx(2.41421342, abs=5e-7)
    # effective near plane of the Vulkan clip rule z_c >= 0: z_eye = -2fn/(f+n)
    z_eye = -p43[3][2] / p43[2][2]
    assert z_eye == pytest.approx(-0.19998, abs=1e-6)


def test_view_matrix(orc):
    # identity camera looks down -Z with view = I (drawer.rs:390-409)
    cam = orc.identity_transforms(1)
    view, _ = orc.camera(cam, 800, 600)
    assert np.array_equal(np.abs(view.reshape(4, 4)), np.eye(4, dtype=np.float32))
    # pitch/yaw argument order of drawer.rs:246: yaw = euler.x, pitch = euler.y
    v = orc.fps_view([1.0, 2.0, 3.0], 0.3, 0.7).reshape(4, 4)  # [col][row]
    cp, sp, cy, sy = math.cos(0.3), math.sin(0.3), math.cos(0.7), math.sin(0.7)
    X, Y, Z = [cy, 0, -sy], [sy * sp, cp, cy * sp], [sy * cp, -sp, cp * cy]
    eye = np.array([1.0, 2.0, 3.0])
    want = np.array([[X[0], Y[0], Z[0], 0], [X[1], Y[1], Z[1], 0], [X[2], Y[2], Z[2], 0],
                     [-np.dot(X, eye), -np.dot(Y, eye), -np.dot(Z, eye), 1]])
    assert np.allclose(v, want, atol=1e-6)
    # Euler::from(Quaternion) round trip away from the gimbal branch
    for yaw, pitch in [(10, 20), (-120, 45), (170, -60), (0, 0), (33, 80)]:
        q = orc.quat_from_euler_deg(yaw, pitch, 0.0)
        e = orc.quat_to_euler(q)
        assert math.degrees(e[0]) == pytest.approx(yaw, abs=2e-3)
        assert math.degrees(e[1]) == pytest.approx(pitch, abs=2e-3)
    # gimbal branch: test > 0.499 -> (0, pi/2, 2 atan2(x, w))
    e = orc.quat_to_euler(orc.quat_from_euler_deg(0.0, 90.0, 0.0))
    assert e[0] == 0.0 and e[1] == pytest.approx(math.pi / 2)


def test_cube_corner_known_answers(orc):
    """Corners of Cube::new under model = T(0,0,-3), view = I, 800x600 (closed forms worked out by hand from the cited formulas)."""
    from renderer_amd import scenes
    s = scenes.cube_scene()
    t = orc.identity_transforms(1)
    orc.translate(t, 0, 0.0, 0.0, -3.0)
    view, proj = orc.camera(orc.identity_transforms(1), 800, 600)
    r = orc.render(s, orc.to_model(t), view, proj, 800, 600, want_clip=True)
    clip = r["clip"]
    # vertex 0 = (-.5,-.5,-.5), vertex 6 = (.5,.5,.5) (cube.rs:30,38)
    assert np.allclose(clip[0], [-0.90533, 1.2071067, 3.3006797, 3.5], rtol=2e-6)
    assert np.allclose(clip[6], [0.90533, -1.2071067, 2.3004797, 2.5], rtol=2e-6)
    ndc0, ndc6 = clip[0][:3] / clip[0][3], clip[6][:3] / clip[6][3]
    assert np.allclose(ndc0, [-0.2586657, 0.34488764, 0.94305134], rtol=2e-6)
    assert np.allclose(ndc6, [0.362132, -0.48284268, 0.9201919], rtol=2e-6)
    px0 = ((ndc0[0] + 1) * 400, (ndc0[1] + 1) * 300)
    assert px0 == pytest.approx((296.53372, 403.4663), abs=2e-3)
    # front face (z = +0.5) is nearest: its depth at the face centre pixel is z_ndc of that plane
    assert r["depth"][300, 400] == pytest.approx(0.9201919, abs=1e-6)
    assert r["prim"][300, 400] in (2, 3)  # front face = triangles 2,3 of the index list (cube.rs:68)


def test_clear_and_srgb_tables(orc):
    # clear (0.7,0.2,0,1) -> sRGB8 (218,124,0,255) (drawer.rs:307 on an sRGB target)
    assert [orc.srgb_encode(x) for x in (0.7, 0.2, 0.0, 1.0)] == [218, 124, 0, 255]
    # decode/encode round trip of every 8-bit code is the identity
    for c in range(256):
        assert orc.srgb_encode(orc.srgb_decode(c)) == c
    # encode is round-to-nearest of the exact transfer function
    for x in np.linspace(0, 1, 5001):
        srgb = 12.92 * x if x <= 0.0031308 else 1.055 * x ** (1 / 2.4) - 0.055
        assert abs(orc.srgb_encode(float(np.float32(x))) - 255 * srgb) <= 0.5 + 1e-3
    assert orc.srgb_encode(float("nan")) == 0 and orc.srgb_encode(-1.0) == 0 and orc.srgb_encode(7.0) == 255


def test_rotation_system_is_deterministic(orc):
    t = orc.identity_transforms(1).repeat(16)
    t = np.ascontiguousarray(t)
    a, b = t.copy(), t.copy()
    for f in range(3):
        orc.rotation_system(a, 3, f, 16.0)
        orc.rotation_system(b, 3, f, 16.0)
    assert a.tobytes() == b.tobytes()
    assert not np.array_equal(a["rotation"], t["rotation"])
    r = [orc.rand01(3, 0, i) for i in range(1000)]
    assert 0.0 <= min(r) and max(r) < 1.0 and 0.4 < np.mean(r) < 0.6
    # Rotation::run: rotate(2u, 10u, 20u) with u = r * (delta*0.01) * 2 (rotation.rs:28-35)
    one = orc.identity_transforms(1)
    u = np.float32(np.float32(np.float32(orc.rand01(3, 0, 0)) * np.float32(np.float32(16.0) * np.float32(0.01))) * np.float32(2.0))
    orc.rotate(one, 0, float(np.float32(2.0) * u), float(np.float32(10.0) * u), float(np.float32(20.0) * u))
    assert np.array_equal(one["rotation"][0], a[0:1].copy()["rotation"][0]) is False  # a advanced 3 frames
    first = orc.identity_transforms(1)
    orc.rotation_system(first, 3, 0, 16.0)
    assert np.array_equal(one["rotation"], first["rotation"])
