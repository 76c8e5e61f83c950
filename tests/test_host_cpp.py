"""The C++ host side (include/sxe_drawer.hpp): the reference's Drawer trait and frame loop
(src/renderer/drawer.rs:26-31, src/main.rs:140-199) driven from C++ through the C ABI."""
import os
import subprocess
import sys

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


def build_jpeg_dump(tmp_path):
    """A small program around sxe_host::load_jpeg: <file.jpg> <out.rgba>, prints width and height."""
    src = tmp_path / "j.cpp"
    src.write_text(
        '#include "sxe_jpeg.hpp"\n#include <iostream>\n'
        'int main(int, char **argv) {\n'
        '    sxe_host::Image im;\n'
        '    const sxe_host::Status s = sxe_host::load_jpeg(argv[1], im);\n'
        '    if (!s) { std::cerr << s.error << "\\n"; return 1; }\n'
        '    std::cout << im.width << " " << im.height << "\\n";\n'
        '    std::ofstream(argv[2], std::ios::binary).write(reinterpret_cast<const char *>(im.rgba.data()), (std::streamsize)im.rgba.size());\n'
        '    return 0;\n}\n')
    exe = str(tmp_path / "j")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", exe,
                           "-L", os.path.join(ROOT, "renderer_amd"), "-lsxe_hip", "-Wl,-rpath," + os.path.join(ROOT, "renderer_amd"),
                           "-Wl,-rpath-link,/opt/rocm/lib"])
    return exe


def build_image_dump(tmp_path):
    """A small program around sxe_host::load_image: <file> <out.rgba>, prints width and height."""
    src = tmp_path / "i.cpp"
    src.write_text(
        '#include "sxe_png.hpp"\n#include <iostream>\n'
        'int main(int, char **argv) {\n'
        '    sxe_host::Image im;\n'
        '    const sxe_host::Status s = sxe_host::load_image(argv[1], im);\n'
        '    if (!s) { std::cerr << s.error << "\\n"; return 1; }\n'
        '    std::cout << im.width << " " << im.height << "\\n";\n'
        '    std::ofstream(argv[2], std::ios::binary).write(reinterpret_cast<const char *>(im.rgba.data()), (std::streamsize)im.rgba.size());\n'
        '    return 0;\n}\n')
    exe = str(tmp_path / "i")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", exe,
                           "-L", os.path.join(ROOT, "renderer_amd"), "-lsxe_hip", "-Wl,-rpath," + os.path.join(ROOT, "renderer_amd"),
                           "-Wl,-rpath-link,/opt/rocm/lib"])
    return exe


def test_png_loader_against_an_independent_decoder(tmp_path):
    """include/sxe_png.hpp: every colour type and bit depth, palette and colour-key transparency,
    interlaced files, stored / fixed / dynamic compression, against an independent decoder, bit for
    bit (PNG is lossless; 16-bit samples keep their high byte); damaged files are refused with a
    message.  load_image picks the decoder by the file's content, like image::open
    (src/renderer/types.rs:161-170)."""
    Image = pytest.importorskip("PIL.Image")
    import struct
    import zlib
    exe = build_image_dump(tmp_path)
    rng = np.random.default_rng(4)
    w, h = 53, 31
    y, x = np.mgrid[0:h, 0:w]
    rgba = np.stack([(x * 5) % 256, (y * 8) % 256, ((x + y) * 3) % 256, np.where((x // 6 + y // 5) % 2, 255, (x * 4) % 256)], -1).astype(np.uint8)
    rgba[::3, ::2, :3] = rng.integers(0, 256, rgba[::3, ::2, :3].shape)  # noise: every filter type gets chosen somewhere
    cases = {}
    cases["rgba8"] = (Image.fromarray(rgba, "RGBA"), {})
    cases["rgb8"] = (Image.fromarray(rgba[..., :3].copy(), "RGB"), {})
    cases["grey8"] = (Image.fromarray(rgba[..., 0].copy(), "L"), {})
    cases["greyalpha8"] = (Image.fromarray(rgba[..., [0, 3]].copy(), "LA"), {})
    cases["grey1"] = (Image.fromarray(rgba[..., 0].copy(), "L").point(lambda v: 255 if v > 127 else 0).convert("1"), {})
    cases["palette8"] = (Image.fromarray(rgba[..., :3].copy(), "RGB").quantize(200), {})
    cases["palette4_transparent"] = (Image.fromarray(rgba[..., :3].copy(), "RGB").quantize(13), {"transparency": bytes([0, 128, 255, 40])})
    cases["rgb8_colour_key"] = (Image.fromarray((rgba[..., :3] // 64 * 64).copy(), "RGB"), {"transparency": (64, 128, 0)})
    cases["grey16"] = (Image.fromarray((rgba[..., 0].astype(np.uint16) * 257 + 13), "I;16"), {})
    cases["rgba8_interlaced_by_hand"] = None  # (written below: PIL does not write Adam7)
    cases["stored"] = (Image.fromarray(rgba, "RGBA"), {"compress_level": 0})
    cases["fixed_codes"] = (Image.fromarray(np.zeros((4, 5, 3), np.uint8) + 7, "RGB"), {"compress_level": 1})
    cases["big_rgb"] = (Image.fromarray(rng.integers(0, 256, (300, 421, 3), dtype=np.uint8), "RGB"), {"compress_level": 9})
    for name, case in cases.items():
        path, out = str(tmp_path / (name + ".png")), str(tmp_path / (name + ".rgba"))
        if case is None:
            # Adam7: the seven reduced images one after the other, each with filter type 0 on every row
            def chunk(kind, body):
                return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body))
            raw = b""
            for x0, y0, dx, dy in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
                sub = rgba[y0::dy, x0::dx]
                if sub.size:
                    raw += b"".join(b"\0" + row.tobytes() for row in sub)
            open(path, "wb").write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 1)) +
                                   chunk(b"IDAT", zlib.compress(raw)[:100]) + chunk(b"IDAT", zlib.compress(raw)[100:]) + chunk(b"IEND", b""))
            want = rgba
        else:
            img, opts = case
            img.save(path, "PNG", **opts)
            ref = Image.open(path)
            if name == "grey16":
                g = (np.asarray(ref).astype(np.uint32) >> 8).astype(np.uint8)
                want = np.stack([g, g, g, np.full_like(g, 255)], -1)
            else:
                want = np.asarray(ref.convert("RGBA"))
        p = subprocess.run([exe, path, out], capture_output=True, text=True)
        hh, ww = want.shape[:2]
        assert p.returncode == 0 and p.stdout.split() == [str(ww), str(hh)], (name, p.stderr)
        got = np.fromfile(out, np.uint8).reshape(hh, ww, 4)
        assert np.array_equal(got, want), (name, int((got != want).sum()))
    assert (np.fromfile(str(tmp_path / "palette4_transparent.rgba"), np.uint8).reshape(h, w, 4)[..., 3] < 255).any()
    assert (np.fromfile(str(tmp_path / "rgb8_colour_key.rgba"), np.uint8).reshape(h, w, 4)[..., 3] == 0).any()
    # a JPEG goes to the JPEG decoder whatever its name; damaged and foreign files are refused
    jpg = str(tmp_path / "really_a_jpeg.png")
    Image.fromarray(rgba[..., :3].copy(), "RGB").save(jpg, "JPEG", quality=90)
    assert subprocess.run([exe, jpg, str(tmp_path / "x.rgba")], capture_output=True, text=True).returncode == 0
    good = bytearray(open(str(tmp_path / "rgba8.png"), "rb").read())
    for name, damage in (("crc", lambda b: b.__setitem__(len(b) // 2, b[len(b) // 2] ^ 0x10)), ("cut", lambda b: b.__delitem__(slice(len(b) - 30, len(b)))),
                         ("text", lambda b: b.__setitem__(slice(0, len(b)), b"hello, this is no image at all"))):
        bad = bytearray(good)
        damage(bad)
        path = str(tmp_path / ("bad_" + name + ".png"))
        open(path, "wb").write(bytes(bad))
        p = subprocess.run([exe, path, str(tmp_path / "x.rgba")], capture_output=True, text=True)
        assert p.returncode == 1 and ("png:" in p.stderr or "neither" in p.stderr), (name, p.stderr)


def test_jpeg_loader_against_an_independent_decoder(tmp_path):
    """include/sxe_jpeg.hpp (the texture-file loader): baseline files in the usual samplings, with
    and without restart markers, decode to what an independent decoder gives, up to the few code
    values by which integer inverse DCTs and chroma interpolation differ; progressive files
    likewise; lossless ones are refused.  The test images are synthetic (the reference's textures are unavailable LFS pointers)."""
    Image = pytest.importorskip("PIL.Image")
    exe = build_jpeg_dump(tmp_path)
    rng = np.random.default_rng(1)

    def picture(w, h):
        y, x = np.mgrid[0:h, 0:w]
        a = np.stack([128 + 100 * np.sin(x / 17.0) * np.cos(y / 11.0), 128 + 90 * np.cos(x / 23.0 + 1), 128 + 80 * np.sin((x + y) / 29.0)], -1)
        return np.clip(a + rng.normal(0, 6, a.shape), 0, 255).astype(np.uint8)

    cases = [("444", dict(subsampling=0, quality=90), (67, 45), 3, 0.1), ("420", dict(subsampling=2, quality=85), (130, 77), 12, 1.5),
             ("422", dict(subsampling=1, quality=75), (64, 64), 12, 1.5), ("grey", dict(quality=80), (50, 33), 3, 0.1),
             ("restart", dict(subsampling=2, quality=85, restart_marker_blocks=3), (100, 60), 12, 1.5),
             ("coarse", dict(subsampling=0, quality=30), (40, 40), 3, 0.1)]
    for name, opts, (w, h), max_diff, mean_diff in cases:
        img = Image.fromarray(picture(w, h))
        if name == "grey":
            img = img.convert("L")
        path, out = str(tmp_path / (name + ".jpg")), str(tmp_path / (name + ".rgba"))
        img.save(path, "JPEG", **opts)
        p = subprocess.run([exe, path, out], capture_output=True, text=True)
        assert p.returncode == 0 and p.stdout.split() == [str(w), str(h)], (name, p.stderr)
        got = np.fromfile(out, np.uint8).reshape(h, w, 4)
        ref = np.asarray(Image.open(path).convert("RGBA"))
        d = np.abs(got.astype(int) - ref.astype(int))
        assert (got[..., 3] == 255).all() and d.max() <= max_diff and d.mean() <= mean_diff, (name, d.max(), d.mean())
    # a grey file whose frame header carries sampling factors 2x2: a one-component scan is not
    # interleaved (T.81 A.2.2), so it holds the same blocks in the same order and decodes alike
    raw = bytearray(open(str(tmp_path / "grey.jpg"), "rb").read())
    sof = raw.find(b"\xff\xc0")
    assert sof > 0 and raw[sof + 9] == 1 and raw[sof + 11] == 0x11
    raw[sof + 11] = 0x22
    open(str(tmp_path / "grey22.jpg"), "wb").write(bytes(raw))
    p = subprocess.run([exe, str(tmp_path / "grey22.jpg"), str(tmp_path / "grey22.rgba")], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    assert open(str(tmp_path / "grey22.rgba"), "rb").read() == open(str(tmp_path / "grey.rgba"), "rb").read()
    # progressive files (spectral selection and successive approximation, T.81 annex G: what the
    # image crate's decoder reads as well), in the same samplings
    for name, opts, (w, h), max_diff, mean_diff in [("p444", dict(subsampling=0, quality=90), (67, 45), 3, 0.1),
                                                    ("p420", dict(subsampling=2, quality=85), (130, 77), 12, 1.5),
                                                    ("p422", dict(subsampling=1, quality=60), (75, 64), 12, 1.5),
                                                    ("pgrey", dict(quality=80), (50, 33), 3, 0.1),
                                                    ("pbig", dict(subsampling=2, quality=95), (301, 203), 12, 1.5)]:
        img = Image.fromarray(picture(w, h))
        if name == "pgrey":
            img = img.convert("L")
        path, out = str(tmp_path / (name + ".jpg")), str(tmp_path / (name + ".rgba"))
        img.save(path, "JPEG", progressive=True, **opts)
        assert b"\xff\xc2" in open(path, "rb").read()[:1000]  # (a progressive frame header)
        p = subprocess.run([exe, path, out], capture_output=True, text=True)
        assert p.returncode == 0 and p.stdout.split() == [str(w), str(h)], (name, p.stderr)
        got = np.fromfile(out, np.uint8).reshape(h, w, 4)
        ref = np.asarray(Image.open(path).convert("RGBA"))
        d = np.abs(got.astype(int) - ref.astype(int))
        assert (got[..., 3] == 255).all() and d.max() <= max_diff and d.mean() <= mean_diff, (name, d.max(), d.mean())
    # a lossless frame header (SOF3) is refused by name
    raw = bytearray(open(str(tmp_path / "444.jpg"), "rb").read())
    raw[raw.find(b"\xff\xc0") + 1] = 0xC3
    (tmp_path / "lossless.jpg").write_bytes(bytes(raw))
    p = subprocess.run([exe, str(tmp_path / "lossless.jpg"), str(tmp_path / "x.rgba")], capture_output=True, text=True)
    assert p.returncode == 1 and "lossless" in p.stderr
    (tmp_path / "junk.jpg").write_bytes(b"\xff\xd8\xff\xdb\x00\x03\x00")
    p = subprocess.run([exe, str(tmp_path / "junk.jpg"), str(tmp_path / "x.rgba")], capture_output=True, text=True)
    assert p.returncode == 1 and "jpeg:" in p.stderr


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


@pytest.mark.gpu
def test_textures_loaded_from_jpeg_files_end_to_end(tmp_path, orc):
    """TextureData::load's path (src/renderer/types.rs:158-188) end to end: the scene file names
    baseline JPEG files, the C++ host loads them (sxe_host::load_jpeg), the frame must equal the
    oracle's frame of the same scene with the texels that loader produced.  Synthetic pictures (the
    reference's own textures are unavailable LFS pointers)."""
    Image = pytest.importorskip("PIL.Image")
    s = scenes.reference_demo_scene(n=64, seed=7, width=320, height=240)
    exe = build_jpeg_dump(tmp_path)
    names, texels = [], []
    for k, t in enumerate(s.textures):
        name = "tex%d.jpg" % k
        Image.fromarray(t[..., :3]).save(str(tmp_path / name), "JPEG", quality=88, subsampling=(0, 2, 1)[k % 3])
        out = str(tmp_path / ("tex%d.rgba" % k))
        p = subprocess.run([exe, str(tmp_path / name), out], capture_output=True, text=True)
        assert p.returncode == 0, p.stderr
        w, h = (int(x) for x in p.stdout.split())
        texels.append(np.fromfile(out, np.uint8).reshape(h, w, 4))
        assert texels[-1].shape == t.shape and not np.array_equal(texels[-1], t)  # lossy, but the same picture
        assert np.abs(texels[-1].astype(int) - t.astype(int)).mean() < 12
        names.append(name)
    p, out, _ = run_demo(tmp_path, s, n_images=2, n_frames=1, texture_files=names)
    assert p.returncode == 0, p.stderr
    w, h = s.size
    got = np.fromfile(out, dtype=np.uint8).reshape(h, w, 4)
    decoded = scenes.Scene(s.vertices, s.indices, s.meshes, s.objects, s.transforms, texels, camera=s.camera, size=s.size)
    view, proj = orc.camera(s.camera, w, h)
    ref = orc.render(decoded, orc.to_model(s.transforms), view, proj, w, h)
    assert np.array_equal(got, ref["color"])
    with_original = orc.render(s, orc.to_model(s.transforms), view, proj, w, h)
    assert not np.array_equal(got, with_original["color"])


@pytest.mark.gpu
def test_textures_loaded_from_png_files_with_alpha_end_to_end(tmp_path, orc):
    """A texture file with an alpha channel (a PNG: image::open(path).to_rgba() keeps the alpha,
    src/renderer/types.rs:161-170) end to end: the C++ host loads it (sxe_host::load_image), the
    frame takes the ordered blending path (BlendState::ALPHA, src/renderer/drawer.rs:660-663) and
    must equal the oracle's frame of the same scene with the same texels.  One of the three textures
    stays a JPEG: the loader goes by content."""
    Image = pytest.importorskip("PIL.Image")
    s = scenes.reference_demo_scene(n=64, seed=7, width=320, height=240)
    s.textures = [s.textures[0], scenes.translucent_texture(128, 5), scenes.translucent_texture(64, 6)]
    names = ["a.jpg", "b.png", "c.png"]
    Image.fromarray(s.textures[0][..., :3]).save(str(tmp_path / names[0]), "JPEG", quality=90)
    Image.fromarray(s.textures[1], "RGBA").save(str(tmp_path / names[1]), "PNG")
    Image.fromarray(s.textures[2], "RGBA").quantize(64).save(str(tmp_path / names[2]), "PNG")  # palette with transparency
    texels = [np.asarray(Image.open(str(tmp_path / n)).convert("RGBA")) for n in names]
    assert np.array_equal(texels[1], s.textures[1]) and (texels[2][..., 3] < 255).any()
    exe = build_jpeg_dump(tmp_path)  # (the JPEG's texels as this build's own decoder gives them)
    p = subprocess.run([exe, str(tmp_path / names[0]), str(tmp_path / "a.rgba")], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    texels[0] = np.fromfile(str(tmp_path / "a.rgba"), np.uint8).reshape(texels[0].shape)
    p, out, _ = run_demo(tmp_path, s, n_images=2, n_frames=1, texture_files=names)
    assert p.returncode == 0, p.stderr
    w, h = s.size
    got = np.fromfile(out, dtype=np.uint8).reshape(h, w, 4)
    decoded = scenes.Scene(s.vertices, s.indices, s.meshes, s.objects, s.transforms, texels, camera=s.camera, size=s.size)
    view, proj = orc.camera(s.camera, w, h)
    ref = orc.render(decoded, orc.to_model(s.transforms), view, proj, w, h)
    assert np.array_equal(got, ref["color"])
    opaque = [t.copy() for t in texels]
    for t in opaque:
        t[..., 3] = 255
    decoded.textures = opaque
    assert not np.array_equal(got, orc.render(decoded, orc.to_model(s.transforms), view, proj, w, h)["color"])  # (the alpha mattered)


def test_damaged_texture_files_under_host_sanitizers():
    """tools/fuzz_image_loaders.py: mutated PNG and JPEG files of every kind the loaders read are
    either decoded or refused with a message, under -fsanitize=address,undefined (host code only)."""
    pytest.importorskip("PIL.Image")
    tool = os.path.join(ROOT, "tools", "fuzz_image_loaders.py")
    run = subprocess.run([sys.executable, tool, "25", "7"], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, (run.stdout[-1500:], run.stderr[-1500:])
    assert "prog420.jpg" in run.stdout and "FAILED" not in run.stdout
