"""The C++ host side (include/sxe_drawer.hpp): the reference's Drawer trait and frame loop
(src/renderer/drawer.rs:26-31, src/main.rs:140-199) driven from C++ through the C ABI."""
import os
import subprocess

import numpy as np
import pytest

from renderer_amd import _build, scenes
from tests.util import write_scene_file

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEMO = os.path.join(ROOT, "host", "engine_demo")


def run_demo(tmp_path, scene, **kw):
    path = str(tmp_path / "scene.sxes")
    out = str(tmp_path / "frame.rgba")
    ppm = str(tmp_path / "frame.ppm")
    write_scene_file(path, scene, **kw)
    p = subprocess.run([DEMO, path, out, ppm], capture_output=True, text=True, timeout=300)
    return p, out, ppm


def test_host_program_is_built_and_links_the_library():
    assert os.path.exists(DEMO), "host/engine_demo is missing: run python -m renderer_amd._build"
    needed = subprocess.run(["readelf", "-d", DEMO], capture_output=True, text=True).stdout
    assert "libsxe_hip.so" in needed
    p = subprocess.run([DEMO], capture_output=True, text=True)
    assert p.returncode == 2 and "usage" in p.stderr


def test_header_compiles_with_warnings_as_errors(tmp_path):
    src = tmp_path / "t.cpp"
    src.write_text('#include "sxe_drawer.hpp"\nint main() { sxe_host::Transform t; t.rotate(1, 2, 3); return t.to_ubo().model[15] == 1.0f ? 0 : 1; }\n')
    exe = str(tmp_path / "t")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", exe,
                           "-L", os.path.join(ROOT, "renderer_amd"), "-lsxe_hip", "-Wl,-rpath," + os.path.join(ROOT, "renderer_amd"),
                           "-Wl,-rpath-link,/opt/rocm/lib"])
    # host-side Transform maths needs no device
    assert subprocess.run([exe]).returncode == 0


def test_png_writer_produces_a_valid_file(tmp_path):
    """write_png (the Presenter stand-in's file output): chunk CRCs, header fields and the pixels
    after inflating the stored-deflate stream."""
    import struct
    import zlib
    src = tmp_path / "p.cpp"
    src.write_text(
        '#include "sxe_drawer.hpp"\n'
        'int main(int, char **argv) {\n'
        '    sxe_host::Image im; im.width = 301; im.height = 77; im.rgba.resize(301 * 77 * 4);\n'
        '    for (size_t i = 0; i < im.rgba.size(); ++i) im.rgba[i] = (uint8_t)((i * 2654435761u) >> 13);\n'
        '    return sxe_host::write_png(argv[1], im).ok() ? 0 : 1;\n'
        '}\n')
    exe, out = str(tmp_path / "p"), str(tmp_path / "p.png")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", exe,
                           "-L", os.path.join(ROOT, "renderer_amd"), "-lsxe_hip", "-Wl,-rpath," + os.path.join(ROOT, "renderer_amd"),
                           "-Wl,-rpath-link,/opt/rocm/lib"])
    subprocess.check_call([exe, out])
    d = open(out, "rb").read()
    assert d[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, []
    while pos < len(d):
        n, = struct.unpack(">I", d[pos:pos + 4])
        kind, body = d[pos + 4:pos + 8], d[pos + 8:pos + 8 + n]
        assert zlib.crc32(kind + body) == struct.unpack(">I", d[pos + 8 + n:pos + 12 + n])[0]
        chunks.append((kind, body))
        pos += 12 + n
    assert [k for k, _ in chunks] == [b"IHDR", b"sRGB", b"IDAT", b"IEND"]
    assert struct.unpack(">IIBBBBB", chunks[0][1]) == (301, 77, 8, 6, 0, 0, 0)
    rows = np.frombuffer(zlib.decompress(chunks[2][1]), np.uint8).reshape(77, 301 * 4 + 1)
    i = np.arange(301 * 77 * 4, dtype=np.uint64)
    assert (rows[:, 0] == 0).all() and np.array_equal(rows[:, 1:], (((i * 2654435761) & 0xFFFFFFFF) >> 13).astype(np.uint8).reshape(77, -1))


@pytest.mark.skipif(__import__("torch").cuda.is_available(), reason="checks the behaviour without a GPU")
def test_host_program_fails_loudly_without_a_gpu(tmp_path):
    p, out, _ = run_demo(tmp_path, scenes.cube_scene(64, 48))
    assert p.returncode == 1 and "sxe error" in p.stderr
    assert not os.path.exists(out)


@pytest.mark.gpu
def test_engine_loop_in_cpp_matches_the_oracle(tmp_path, orc):
    """64 cubes with three textures (the world of src/main.rs:201-227), four frames of the loop with
    a fixed rotation per frame; the frame presented last must equal the oracle's."""
    s = scenes.reference_demo_scene(n=64, seed=7, width=320, height=240)
    rot = (2.0, 10.0, 20.0)
    p, out, ppm = run_demo(tmp_path, s, n_images=2, n_frames=4, rotate_deg=rot)
    assert p.returncode == 0, p.stderr
    w, h = s.size
    got = np.fromfile(out, dtype=np.uint8).reshape(h, w, 4)
    t = s.transforms.copy()
    for _ in range(4):
        for i in range(len(t)):
            orc.rotate(t, i, *rot)
    view, proj = orc.camera(s.camera, w, h)
    ref = orc.render(s, orc.to_model(t), view, proj, w, h)
    assert np.abs(got.astype(np.int16) - ref["color"].astype(np.int16)).max() <= 1
    assert np.array_equal(got, ref["color"])
    head = open(ppm, "rb").read(15)
    assert head.startswith(b"P6\n320 240\n255\n")
    assert os.path.getsize(ppm) == 15 + w * h * 3


@pytest.mark.gpu
def test_cpp_drawer_reports_a_missing_texture_image(tmp_path):
    """A drawable without a texture under the textured shader is an error result, not a crash."""
    s = scenes.cube_scene(64, 48)
    s.objects["texture"] = -1
    p, out, _ = run_demo(tmp_path, s)
    assert p.returncode == 1 and "update_drawables" in p.stderr and "texture" in p.stderr
