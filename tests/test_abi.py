"""CPU-side checks of the C ABI: the library loads, exports what include/sxe.h declares, its host
maths equals the oracle bit for bit, and it refuses to render without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from renderer_amd import api, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "sxe.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sxe_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = api.lib()
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), "libsxe_hip.so does not export %s" % n
    assert sorted(api.EXPORTS) == names
    assert b"gfx950" in L.sxe_version()


def test_struct_layouts_match_the_reference_wire_formats():
    # Vertex: 32 B with offsets 0/12/24 (drawer.rs:665-696); Transform: 40 B (transform.rs:16-21)
    assert api.VERTEX.itemsize == 32 and api.VERTEX.fields["color"][1] == 12 and api.VERTEX.fields["uv"][1] == 24
    assert api.TRANSFORM.itemsize == 40 == C.sizeof(api.TransformC)
    assert C.sizeof(api.Config) == 60
    assert C.sizeof(api.DrawableC) == 80


def test_host_maths_equals_the_oracle_bitwise(orc):
    rng = np.random.default_rng(3)
    n = 200
    a = api.identity_transforms(n)
    b = orc.identity_transforms(n)
    assert a.tobytes() == b.tobytes()
    pos = rng.uniform(-50, 50, (n, 3)).astype(np.float32)
    eul = rng.uniform(-400, 400, (n, 3)).astype(np.float32)
    a["scale"] = b["scale"] = rng.uniform(0.1, 4, (n, 3)).astype(np.float32)
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    for i in range(n):
        api.translate(a, i, *map(float, pos[i]))
        orc.translate(b, i, *map(float, pos[i]))
        for _ in range(3):
            api.rotate(a, i, *map(float, eul[i]))
            orc.rotate(b, i, *map(float, eul[i]))
    assert a.tobytes() == b.tobytes()
    assert api.to_model(a).tobytes() == orc.to_model(b).tobytes()
    for i in range(0, n, 5):
        for w, h in [(800, 600), (3840, 2160), (512, 512)]:
            va, pa = api.camera_matrices(a[i:i + 1], w, h)
            vb, pb = orc.camera(b[i:i + 1], w, h)
            assert va.tobytes() == vb.tobytes() and pa.tobytes() == pb.tobytes()


def test_orbit_cameras_look_at_their_target(orc):
    cams = scenes.orbit_cameras(32, 4, radius=60.0, target=(0.0, 0.0, -22.5))
    for i in range(len(cams)):
        view, _ = api.camera_matrices(cams[i:i + 1], 512, 512)
        v = view.reshape(4, 4).T.astype(np.float64)
        p = v @ np.array([0.0, 0.0, -22.5, 1.0])
        # the target sits on the camera's -Z axis at the orbit radius
        assert abs(p[0]) < 2e-3 and abs(p[1]) < 2e-3 and p[2] == pytest.approx(-60.0, abs=2e-3)


def test_create_fails_loudly_without_a_gpu(hip_available):
    if hip_available:
        pytest.skip("a GPU is present; the no-device refusal is exercised on CPU-only hosts")
    with pytest.raises(api.SxeError) as e:
        api.HipDrawer(64, 64)
    assert e.value.code == api.E_DEVICE and "no CPU path" in str(e.value)


def test_create_rejects_bad_configs():
    L = api.lib()
    ctx = C.c_void_p()
    cfg = api.Config()
    L.sxe_default_config(C.byref(cfg), 0, 10)
    assert L.sxe_create(C.byref(cfg), C.byref(ctx)) == api.E_INVALID and not ctx.value
    assert b"frame size" in L.sxe_last_error(None)
    L.sxe_default_config(C.byref(cfg), 9000, 10)
    assert L.sxe_create(C.byref(cfg), C.byref(ctx)) == api.E_INVALID
    L.sxe_default_config(C.byref(cfg), 64, 64)
    assert (cfg.fov_deg, cfg.n_images) == (45.0, 2)
    assert list(cfg.clear) == pytest.approx([0.7, 0.2, 0.0, 1.0])
    cfg.shading = 9
    assert L.sxe_create(C.byref(cfg), C.byref(ctx)) == api.E_INVALID
    cfg.shading = 0
    cfg.scissor_y0, cfg.scissor_y1 = 40, 100
    assert L.sxe_create(C.byref(cfg), C.byref(ctx)) == api.E_INVALID
    assert L.sxe_create(None, C.byref(ctx)) == api.E_INVALID
    L.sxe_destroy(None)  # harmless


def test_scene_builders():
    v, idx = scenes.cube_mesh()
    assert len(v) == 24 and len(idx) == 36 and idx.max() == 23
    assert np.abs(v["pos"]).max() == 0.5 and set(np.unique(v["uv"])) == {0.0, 1.0}
    assert idx[:6].tolist() == [0, 1, 2, 2, 3, 0]
    v, idx = scenes.grid_sphere_mesh()
    assert len(v) == 561 and len(idx) == 3 * 1024
    s = scenes.config_scene(3, scale=0.01)
    assert len(s.objects) == 100 and s.n_triangles == 102400
    for t in s.textures:
        assert (t[..., 3] == 255).all()
    s1 = scenes.config_scene(3, scale=0.01)
    assert s1.transforms.tobytes() == s.transforms.tobytes()
    d = s.drawables()
    assert len(d) == 100 and len(d[0].vertices) == 561


def test_srgb_encode_table_equals_the_threshold_definition(orc):
    """The tile kernel encodes linear colour with a bucket table over the float bits.  Its host
    twin (same table builder, same lookup) must agree with the oracle's encoder -- the number of
    thresholds <= x -- on every float next to a code boundary and on a large random sample."""
    L = api.lib()
    L.sxe_srgb_encode_host.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    xs = []
    # code boundaries: bisect the oracle's encoder between 0 and 1 for every code step
    for code in range(1, 256):
        lo, hi = np.float32(0.0).view(np.uint32), np.float32(1.0).view(np.uint32)
        lo, hi = int(lo), int(hi)
        while hi - lo > 1:  # positive floats order like their bit patterns
            mid = (lo + hi) // 2
            if orc.srgb_encode(float(np.uint32(mid).view(np.float32))) >= code:
                hi = mid
            else:
                lo = mid
        xs.extend(range(max(hi - 3, 0), hi + 4))
    near = np.array(xs, dtype=np.uint32).view(np.float32)
    rng = np.random.default_rng(5)
    sample = np.concatenate([
        near,
        rng.uniform(0.0, 1.0, 200000).astype(np.float32),
        (rng.uniform(0.0, 1.0, 50000) ** 4).astype(np.float32),          # dense near zero
        np.array([0.0, -0.0, 1.0, 1.0000001, 2.0, 1e-30, 1e-45, -1e-3, 0.00012207031, 0.0031308, 0.5], dtype=np.float32),
    ])
    sample = np.ascontiguousarray(sample)
    got = np.zeros(len(sample), dtype=np.uint8)
    assert L.sxe_srgb_encode_host(sample.ctypes.data, got.ctypes.data, len(sample)) == 0
    want = np.array([orc.srgb_encode(float(v)) for v in sample], dtype=np.uint8)
    bad = np.nonzero(got != want)[0]
    assert len(bad) == 0, "x=%r: table %d, definition %d" % (sample[bad[0]], got[bad[0]], want[bad[0]])
    assert got[len(near) - 1] == 255 and set(np.unique(got[:len(near)])) == set(range(256))


def test_rust_declarations_cover_the_header():
    """bindings/rust/sxe-sys/src/lib.rs cannot be compiled here (no Rust toolchain); at least it must
    declare every function of include/sxe.h with the same number of arguments, and its #[repr(C)]
    structs must add up to the sizes of the C ones."""
    rs = open(os.path.join(ROOT, "bindings", "rust", "sxe-sys", "src", "lib.rs")).read()
    header = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "sxe.h")).read(), flags=re.S)
    c_fns = {m.group(1): m.group(2) for m in re.finditer(r"\b(sxe_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", header)}
    rs_fns = {m.group(1): m.group(2) for m in re.finditer(r"pub fn (sxe_[a-z0-9_]+)\(([^)]*)\)", rs)}
    assert sorted(c_fns) == sorted(rs_fns) == declared_symbols()

    def n_args(text):
        text = text.strip()
        return 0 if text in ("", "void") else text.count(",") + 1
    for name in c_fns:
        assert n_args(c_fns[name]) == n_args(rs_fns[name]), name
    size = {"u8": 1, "i32": 4, "u32": 4, "f32": 4, "u64": 8, "*": 8, "sxe_transform": 40, "sxe_vertex": 32}
    align = {"u8": 1, "i32": 4, "u32": 4, "f32": 4, "u64": 8, "*": 8, "sxe_transform": 4, "sxe_vertex": 4}
    want = {"sxe_vertex": 32, "sxe_transform": 40, "sxe_config": C.sizeof(api.Config), "sxe_drawable": C.sizeof(api.DrawableC),
            "sxe_timings": C.sizeof(api.Timings), "sxe_frame_stats": C.sizeof(api.FrameStats), "sxe_texture_desc": 16,
            "sxe_mesh_desc": 32, "sxe_instance": 52, "sxe_light": C.sizeof(api.Light)}
    for name, expect in want.items():
        body = re.search(r"pub struct %s \{(.*?)\n\}" % name, rs, re.S).group(1)
        off, biggest = 0, 1
        for ty in re.findall(r"pub \w+: ([^,\n]+),", body):
            ty = ty.strip()
            m = re.match(r"\[(\w+); (\d+)\]", ty)
            base, count = (m.group(1), int(m.group(2))) if m else (("*" if ty.startswith("*") else ty), 1)
            a = align[base]
            off = (off + a - 1) // a * a + size[base] * count
            biggest = max(biggest, a)
        assert (off + biggest - 1) // biggest * biggest == expect, name
