#!/usr/bin/env python3
"""Development (CPU only): damaged texture files against include/sxe_png.hpp and include/sxe_jpeg.hpp
under the host sanitizers.  Valid PNG and JPEG files of every kind the loaders read are mutated
(bytes flipped, runs zeroed, truncated, chunk lengths and header fields changed with the PNG check
sums made right again, so that the damage reaches the inflater and the filters) and loaded by a
small program built with -fsanitize=address,undefined: a file is either decoded to width x height x 4
bytes or refused with a message; anything else (a sanitizer report, a crash, a time limit) fails.

  fuzz_image_loaders.py [cases per seed file] [seed]
"""
import io, os, struct, subprocess, sys, tempfile, zlib
import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROGRAM = r'''
#include "sxe_png.hpp"
#include <iostream>
int main(int argc, char **argv) {
    int refused = 0;
    for (int k = 1; k < argc; ++k) {
        sxe_host::Image im;
        const sxe_host::Status s = sxe_host::load_image(argv[k], im);
        if (!s) { ++refused; continue; }
        if (im.rgba.size() != (size_t)im.width * im.height * 4) { std::cerr << "size mismatch " << argv[k] << "\n"; return 2; }
    }
    std::cout << refused << " refused of " << (argc - 1) << "\n";
    return 0;
}
'''


def seeds(rng):
    w, h = 37, 23
    y, x = np.mgrid[0:h, 0:w]
    rgba = np.stack([(x * 7) % 256, (y * 11) % 256, ((x + y) * 5) % 256, np.where((x // 4 + y // 3) % 2, 255, (x * 6) % 256)], -1).astype(np.uint8)
    rgba[::2, ::3, :3] = rng.integers(0, 256, rgba[::2, ::3, :3].shape)
    out = {}

    def png(name, im, **kw):
        b = io.BytesIO(); im.save(b, "PNG", **kw); out[name + ".png"] = b.getvalue()

    def jpg(name, im, **kw):
        b = io.BytesIO(); im.save(b, "JPEG", **kw); out[name + ".jpg"] = b.getvalue()
    png("rgba8", Image.fromarray(rgba, "RGBA"))
    png("rgb8", Image.fromarray(rgba[..., :3].copy(), "RGB"))
    png("grey8", Image.fromarray(rgba[..., 0].copy(), "L"))
    png("la8", Image.fromarray(rgba[..., [0, 3]].copy(), "LA"))
    png("pal", Image.fromarray(rgba[..., :3].copy(), "RGB").quantize(60))
    png("bit1", Image.fromarray(rgba[..., 0].copy(), "L").point(lambda v: 255 if v > 127 else 0).convert("1"))
    png("stored", Image.fromarray(rgba, "RGBA"), compress_level=0)
    out["adam7.png"] = interlaced_png(rgba)
    # kinds the imaging library does not write: 16-bit samples, a palette with transparency, a colour key
    wide = (rgba[..., :3].astype(np.uint16) * 257 + 3).astype(">u2")
    out["rgb16.png"] = plain_png(w, h, 16, 2, [row.tobytes() for row in wide])
    out["grey16.png"] = plain_png(w, h, 16, 0, [row.tobytes() for row in wide[..., 0]])
    out["paltrns.png"] = plain_png(w, h, 8, 3, [(row % 7).astype(np.uint8).tobytes() for row in rgba[..., 0]],
                                   [(b"PLTE", bytes(range(21))), (b"tRNS", bytes([0, 128, 255, 7]))])
    out["greykey.png"] = plain_png(w, h, 8, 0, [row.tobytes() for row in rgba[..., 1]], [(b"tRNS", struct.pack(">H", 33))])
    rgb = Image.fromarray(rgba[..., :3].copy(), "RGB")
    jpg("base444", rgb, quality=90, subsampling=0)
    jpg("base420", rgb, quality=75, subsampling=2)
    jpg("grey", Image.fromarray(rgba[..., 0].copy(), "L"), quality=80)
    jpg("prog444", rgb, quality=90, subsampling=0, progressive=True)
    jpg("prog420", rgb, quality=70, subsampling=2, progressive=True)
    return out


def chunk(kind, data):
    return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data) & 0xFFFFFFFF)


def interlaced_png(rgba):
    """An Adam7 file written by hand (the imaging library writes none)."""
    h, w = rgba.shape[:2]
    raw = b""
    for x0, y0, dx, dy in [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]:
        sub = rgba[y0::dy, x0::dx]
        if sub.size == 0:
            continue
        for row in sub:
            raw += b"\x00" + row.tobytes()
    ihdr = struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 1)
    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", ihdr) + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b"")


def plain_png(w, h, depth, colour_type, rows, extra=()):
    raw = b"".join(b"\x00" + r for r in rows)
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, colour_type, 0, 0, 0)) +
            b"".join(chunk(k, d) for k, d in extra) + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))


def png_chunks(data):
    pos, out = 8, []
    while pos + 12 <= len(data):
        n = struct.unpack(">I", data[pos:pos + 4])[0]
        out.append((data[pos + 4:pos + 8], bytearray(data[pos + 8:pos + 8 + n])))
        pos += 12 + n
    return out


def mutate(name, data, rng):
    b = bytearray(data)
    kind = int(rng.integers(0, 6))
    if name.endswith(".png") and kind >= 3:
        # damage inside the chunks, check sums made right again
        chunks = png_chunks(data)
        which = int(rng.integers(0, len(chunks)))
        k, d = chunks[which]
        if k == b"IDAT" and kind == 5 and len(d) > 8:
            # damage under the compression: the filter bytes and samples themselves
            try:
                raw = bytearray(zlib.decompress(bytes(d)))
                for _ in range(int(rng.integers(1, 8))):
                    raw[int(rng.integers(0, len(raw)))] = int(rng.integers(0, 256))
                if rng.random() < 0.3:
                    raw = raw[:int(rng.integers(0, len(raw)))]
                d = bytearray(zlib.compress(bytes(raw)))
            except zlib.error:
                pass
        elif len(d):
            for _ in range(int(rng.integers(1, 6))):
                d[int(rng.integers(0, len(d)))] = int(rng.integers(0, 256))
        chunks[which] = (k, d)
        if rng.random() < 0.1:
            chunks.pop(int(rng.integers(0, len(chunks))))
        return b"\x89PNG\r\n\x1a\n" + b"".join(chunk(k, bytes(d)) for k, d in chunks)
    if kind == 0:
        for _ in range(int(rng.integers(1, 10))):
            b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
    elif kind == 1:
        at = int(rng.integers(0, len(b))); n = int(rng.integers(1, 64))
        b[at:at + n] = bytes(int(rng.integers(0, 256)) for _ in range(min(n, len(b) - at)))
    elif kind == 2:
        b = b[:int(rng.integers(0, len(b)))]
    else:
        for _ in range(int(rng.integers(1, 4))):
            at = int(rng.integers(0, max(len(b) - 2, 1)))
            b[at:at + 2] = struct.pack(">H", int(rng.integers(0, 65536)))
    return bytes(b)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    with tempfile.TemporaryDirectory() as tmp:
        src = os.path.join(tmp, "i.cpp"); exe = os.path.join(tmp, "i")
        open(src, "w").write(PROGRAM)
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-static-libasan",
                               "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
        env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:verify_asan_link_order=0")
        total = refused = 0
        for name, data in seeds(rng).items():
            paths = []
            whole = os.path.join(tmp, "ok_" + name)
            open(whole, "wb").write(data)
            r = subprocess.run([exe, whole], env=env, capture_output=True, text=True, timeout=60)
            assert r.returncode == 0 and r.stdout.startswith("0 refused"), (name, r.stdout, r.stderr[-2000:])
            for c in range(cases):
                p = os.path.join(tmp, "%d_%s" % (c, name))
                open(p, "wb").write(mutate(name, data, rng))
                paths.append(p)
            for at in range(0, len(paths), 100):
                r = subprocess.run([exe] + paths[at:at + 100], env=env, capture_output=True, text=True, timeout=120)
                if r.returncode != 0:
                    print("FAILED on a file among", paths[at], "...:", r.stderr[-3000:])
                    for p in paths[at:at + 100]:
                        q = subprocess.run([exe, p], env=env, capture_output=True, text=True, timeout=60)
                        if q.returncode != 0:
                            keep = os.path.join(ROOT, "gpurun_out", "bad_" + os.path.basename(p))
                            os.makedirs(os.path.dirname(keep), exist_ok=True)
                            open(keep, "wb").write(open(p, "rb").read())
                            print("kept", keep)
                            break
                    return 1
                refused += int(r.stdout.split()[0]); total += len(paths[at:at + 100])
            for p in paths:
                os.remove(p)
            print("%-14s %d damaged files so far, %d refused, the others decoded" % (name, total, refused), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
