// sxe_png.hpp -- PNG -> RGBA8 loader for texture files (C++17, header-only, host only), and
// load_image(), which opens a texture file whatever it holds.
//
// The reference reads its textures with `image::open(path)?.to_rgba()` (src/renderer/types.rs:158-188):
// the image crate picks the decoder by the file's content, so a PNG works there as well as the
// JPEGs the reference ships, and a PNG's alpha channel reaches the blend stage
// (BlendState::ALPHA, src/renderer/drawer.rs:660-663).  This is the stand-in for that call for PNG
// files: all five colour types, bit depths 1 to 16 (16-bit samples keep their high byte), palette
// and colour-key transparency (tRNS), Adam7 interlacing; every chunk's CRC and the stream's
// Adler-32 are checked.  Ancillary chunks (gamma, profiles, text) are skipped: the texels are
// taken as sRGB code values, as the reference uploads them (Rgba8Srgb, src/renderer/types.rs:186).
// The reference's own texture files are not available (LFS pointers) and are JPEGs anyway: there
// is no reference output to compare with (parity unpinned); the tests compare with another decoder.
#ifndef SXE_PNG_HPP
#define SXE_PNG_HPP

#include "sxe_jpeg.hpp"

#include <cstdlib>

namespace sxe_host {
namespace png_detail {

inline uint32_t crc32(const uint8_t *p, size_t n, uint32_t crc = 0)
{
    static uint32_t table[256];
    static bool ready = false;
    if (!ready) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        ready = true;
    }
    crc = ~crc;
    for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 255u] ^ (crc >> 8);
    return ~crc;
}

// DEFLATE (RFC 1951) inside a zlib stream (RFC 1950).
struct Inflater {
    const uint8_t *p, *end;
    uint32_t acc = 0;
    int n = 0;
    bool short_input = false;
    Inflater(const uint8_t *b, const uint8_t *e) : p(b), end(e) {}
    uint32_t bits(int k)
    {
        while (n < k) {
            if (p >= end) { short_input = true; return 0; }
            acc |= (uint32_t)*p++ << n;
            n += 8;
        }
        const uint32_t v = acc & ((k < 32) ? ((1u << k) - 1u) : ~0u);
        acc >>= k;
        n -= k;
        return v;
    }
    struct Code { // canonical Huffman code: count of codes per length, symbols in code order
        uint16_t count[16] = {}, symbol[288] = {};
        void build(const uint8_t *lengths, int n_sym)
        {
            for (auto &c : count) c = 0;
            for (int i = 0; i < n_sym; ++i) count[lengths[i]]++;
            count[0] = 0;
            uint16_t offs[16] = {};
            for (int l = 1; l < 16; ++l) offs[l] = (uint16_t)(offs[l - 1] + count[l - 1]);
            for (int i = 0; i < n_sym; ++i)
                if (lengths[i]) symbol[offs[lengths[i]]++] = (uint16_t)i;
        }
    };
    int decode(const Code &c)
    {
        int code = 0, first = 0, index = 0;
        for (int l = 1; l < 16; ++l) {
            code |= (int)bits(1);
            if (short_input) return -1;
            const int cnt = c.count[l];
            if (code - cnt < first) return c.symbol[index + (code - first)];
            index += cnt;
            first += cnt;
            first <<= 1;
            code <<= 1;
        }
        return -1;
    }
    const char *run(std::vector<uint8_t> &out, size_t expect)
    {
        static const uint16_t len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
        static const uint8_t len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
        static const uint16_t dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
        static const uint8_t dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
        if (end - p < 6) return "zlib stream too short";
        const int cmf = p[0], flg = p[1];
        if ((cmf & 15) != 8 || (cmf >> 4) > 7 || ((cmf << 8) | flg) % 31 != 0 || (flg & 32)) return "bad zlib header";
        p += 2;
        out.clear();
        out.reserve(expect);
        for (bool last = false; !last;) {
            last = bits(1) != 0;
            const uint32_t type = bits(2);
            if (short_input) return "compressed data ends early";
            if (type == 0) {
                acc = 0;
                n = 0; // to the next byte
                if (end - p < 4) return "compressed data ends early";
                const uint32_t len = p[0] | (p[1] << 8), nlen = p[2] | (p[3] << 8);
                p += 4;
                if ((len ^ 0xFFFFu) != nlen || (size_t)(end - p) < len) return "bad stored block";
                out.insert(out.end(), p, p + len);
                p += len;
            } else if (type == 1 || type == 2) {
                Code lit, dist;
                uint8_t lengths[320] = {};
                if (type == 1) {
                    for (int i = 0; i < 144; ++i) lengths[i] = 8;
                    for (int i = 144; i < 256; ++i) lengths[i] = 9;
                    for (int i = 256; i < 280; ++i) lengths[i] = 7;
                    for (int i = 280; i < 288; ++i) lengths[i] = 8;
                    lit.build(lengths, 288);
                    for (int i = 0; i < 30; ++i) lengths[i] = 5;
                    dist.build(lengths, 30);
                } else {
                    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                    const int hlit = (int)bits(5) + 257, hdist = (int)bits(5) + 1, hclen = (int)bits(4) + 4;
                    if (hlit > 286 || hdist > 30) return "bad block header";
                    uint8_t cl[19] = {};
                    for (int i = 0; i < hclen; ++i) cl[order[i]] = (uint8_t)bits(3);
                    Code clc;
                    clc.build(cl, 19);
                    int at = 0;
                    while (at < hlit + hdist) {
                        const int s = decode(clc);
                        if (s < 0) return "bad code lengths";
                        if (s < 16) {
                            lengths[at++] = (uint8_t)s;
                        } else {
                            int rep, val = 0;
                            if (s == 16) {
                                if (at == 0) return "bad code lengths";
                                val = lengths[at - 1];
                                rep = 3 + (int)bits(2);
                            } else if (s == 17) {
                                rep = 3 + (int)bits(3);
                            } else {
                                rep = 11 + (int)bits(7);
                            }
                            if (at + rep > hlit + hdist) return "bad code lengths";
                            while (rep--) lengths[at++] = (uint8_t)val;
                        }
                    }
                    if (short_input) return "compressed data ends early";
                    lit.build(lengths, hlit);
                    dist.build(lengths + hlit, hdist);
                }
                for (;;) {
                    const int s = decode(lit);
                    if (s < 0) return short_input ? "compressed data ends early" : "bad literal/length code";
                    if (s < 256) {
                        out.push_back((uint8_t)s);
                    } else if (s == 256) {
                        break;
                    } else {
                        if (s > 285) return "bad length code";
                        const int len = len_base[s - 257] + (int)bits(len_extra[s - 257]);
                        const int ds = decode(dist);
                        if (ds < 0 || ds > 29) return "bad distance code";
                        const size_t d = dist_base[ds] + bits(dist_extra[ds]);
                        if (short_input) return "compressed data ends early";
                        if (d > out.size()) return "distance reaches before the start of the data";
                        const size_t from = out.size() - d;
                        for (int i = 0; i < len; ++i) out.push_back(out[from + (size_t)i]);
                    }
                    if (out.size() > expect + 8) return "more image data than the header announces";
                }
            } else {
                return "bad block type";
            }
        }
        // Adler-32 of the uncompressed data follows, on a byte boundary
        acc = 0;
        n = 0;
        if (end - p < 4) return "zlib checksum missing";
        uint32_t a = 1, b = 0;
        for (uint8_t v : out) {
            a = (a + v) % 65521u;
            b = (b + a) % 65521u;
        }
        const uint32_t want = ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
        if (want != ((b << 16) | a)) return "zlib checksum mismatch";
        return nullptr;
    }
};

inline int paeth(int a, int b, int c)
{
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

} // namespace png_detail

inline Status decode_png(const uint8_t *data, size_t size, Image &out)
{
    using namespace png_detail;
    auto err = [](const std::string &what) { return Status::Err("png: " + what); };
    static const uint8_t magic[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (size < 8 || memcmp(data, magic, 8) != 0) return err("not a PNG file");
    auto be32 = [](const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; };
    uint32_t width = 0, height = 0;
    int depth = 0, ctype = -1, interlace = 0;
    std::vector<uint8_t> idat, palette, trns;
    bool seen_end = false;
    for (size_t at = 8; at + 12 <= size && !seen_end;) {
        const uint32_t len = be32(data + at);
        if (len > size - at - 12) return err("chunk runs past the end of the file");
        const uint8_t *type = data + at + 4, *body = data + at + 8;
        if (crc32(type, 4 + (size_t)len) != be32(body + len)) return err("chunk checksum mismatch");
        const std::string name(reinterpret_cast<const char *>(type), 4);
        if (name == "IHDR") {
            if (len != 13) return err("bad header chunk");
            width = be32(body);
            height = be32(body + 4);
            depth = body[8];
            ctype = body[9];
            interlace = body[12];
            if (body[10] != 0 || body[11] != 0 || interlace > 1) return err("unknown compression, filter or interlace method");
        } else if (name == "PLTE") {
            palette.assign(body, body + len);
        } else if (name == "tRNS") {
            trns.assign(body, body + len);
        } else if (name == "IDAT") {
            idat.insert(idat.end(), body, body + len);
        } else if (name == "IEND") {
            seen_end = true;
        } else if (!(type[0] & 0x20)) {
            return err("unknown critical chunk '" + name + "'");
        }
        at += 12 + (size_t)len;
    }
    if (ctype < 0) return err("no header chunk");
    if (!seen_end) return err("no end chunk");
    if (width == 0 || height == 0 || width > 16384 || height > 16384) return err("image size outside 1..16384");
    int channels;
    switch (ctype) {
    case 0: channels = 1; break;
    case 2: channels = 3; break;
    case 3: channels = 1; break;
    case 4: channels = 2; break;
    case 6: channels = 4; break;
    default: return err("unknown colour type");
    }
    const bool depth_ok = ctype == 0 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)
                        : ctype == 3 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8) : (depth == 8 || depth == 16);
    if (!depth_ok) return err("bit depth not allowed for the colour type");
    if (ctype == 3 && (palette.empty() || palette.size() % 3)) return err("palette missing or malformed");
    const int bits_per_pixel = channels * depth, bpp = std::max(1, bits_per_pixel / 8); // bytes per whole pixel (filter unit)

    // the passes: one for a plain image, seven for Adam7
    struct Pass { uint32_t x0, y0, dx, dy; };
    static const Pass adam7[7] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
    static const Pass plain[1] = {{0, 0, 1, 1}};
    const Pass *passes = interlace ? adam7 : plain;
    const int n_passes = interlace ? 7 : 1;
    size_t expect = 0;
    for (int k = 0; k < n_passes; ++k) {
        const uint32_t pw = (width - passes[k].x0 + passes[k].dx - 1) / passes[k].dx, ph = (height - passes[k].y0 + passes[k].dy - 1) / passes[k].dy;
        if (passes[k].x0 < width && passes[k].y0 < height && pw && ph) expect += (size_t)ph * (1 + ((size_t)pw * bits_per_pixel + 7) / 8);
    }
    std::vector<uint8_t> raw;
    {
        Inflater z(idat.data(), idat.data() + idat.size());
        if (const char *what = z.run(raw, expect)) return err(what);
    }
    if (raw.size() != expect) return err("image data does not match the header's size");

    out.width = width;
    out.height = height;
    out.rgba.assign((size_t)width * height * 4, 255);
    // colour key of grey and RGB images (tRNS): one sample value per channel, 16 bits each
    int key[3] = {-1, -1, -1};
    if (ctype == 0 && trns.size() >= 2) key[0] = (trns[0] << 8) | trns[1];
    if (ctype == 2 && trns.size() >= 6)
        for (int c = 0; c < 3; ++c) key[c] = (trns[2 * c] << 8) | trns[2 * c + 1];
    size_t at = 0;
    std::vector<uint8_t> prev, line;
    for (int k = 0; k < n_passes; ++k) {
        const Pass &ps = passes[k];
        if (ps.x0 >= width || ps.y0 >= height) continue;
        const uint32_t pw = (width - ps.x0 + ps.dx - 1) / ps.dx, ph = (height - ps.y0 + ps.dy - 1) / ps.dy;
        const size_t row_bytes = ((size_t)pw * bits_per_pixel + 7) / 8;
        prev.assign(row_bytes, 0);
        line.resize(row_bytes);
        for (uint32_t j = 0; j < ph; ++j) {
            const int filter = raw[at++];
            if (filter > 4) return err("unknown filter type");
            const uint8_t *src = raw.data() + at;
            at += row_bytes;
            for (size_t i = 0; i < row_bytes; ++i) {
                const int a = i >= (size_t)bpp ? line[i - bpp] : 0, b = prev[i], c = i >= (size_t)bpp ? prev[i - bpp] : 0;
                const int pred = filter == 0 ? 0 : filter == 1 ? a : filter == 2 ? b : filter == 3 ? (a + b) / 2 : paeth(a, b, c);
                line[i] = (uint8_t)(src[i] + pred);
            }
            const uint32_t y = ps.y0 + j * ps.dy;
            for (uint32_t i = 0; i < pw; ++i) {
                // the pixel's samples at full precision (for the colour key) and as 8-bit values
                int full[4] = {0, 0, 0, 0}, v8[4] = {0, 0, 0, 255};
                for (int c = 0; c < channels; ++c) {
                    if (depth == 16) {
                        const size_t o = ((size_t)i * channels + c) * 2;
                        full[c] = (line[o] << 8) | line[o + 1];
                        v8[c] = line[o]; // the high byte
                    } else if (depth == 8) {
                        full[c] = v8[c] = line[(size_t)i * channels + c];
                    } else { // 1, 2 or 4 bits, one channel, most significant bits first
                        const size_t bit = (size_t)i * depth;
                        full[c] = (line[bit / 8] >> (8 - depth - (int)(bit % 8))) & ((1 << depth) - 1);
                        v8[c] = ctype == 3 ? full[c] : full[c] * 255 / ((1 << depth) - 1);
                    }
                }
                uint8_t *px = &out.rgba[((size_t)y * width + ps.x0 + (size_t)i * ps.dx) * 4];
                switch (ctype) {
                case 0:
                    px[0] = px[1] = px[2] = (uint8_t)v8[0];
                    px[3] = full[0] == key[0] ? 0 : 255;
                    break;
                case 2:
                    px[0] = (uint8_t)v8[0]; px[1] = (uint8_t)v8[1]; px[2] = (uint8_t)v8[2];
                    px[3] = (full[0] == key[0] && full[1] == key[1] && full[2] == key[2]) ? 0 : 255;
                    break;
                case 3: {
                    const size_t e = (size_t)full[0];
                    if (3 * e + 2 >= palette.size()) return err("palette index out of range");
                    px[0] = palette[3 * e]; px[1] = palette[3 * e + 1]; px[2] = palette[3 * e + 2];
                    px[3] = e < trns.size() ? trns[e] : 255;
                    break;
                }
                case 4:
                    px[0] = px[1] = px[2] = (uint8_t)v8[0];
                    px[3] = (uint8_t)v8[1];
                    break;
                default:
                    px[0] = (uint8_t)v8[0]; px[1] = (uint8_t)v8[1]; px[2] = (uint8_t)v8[2]; px[3] = (uint8_t)v8[3];
                    break;
                }
            }
            prev.swap(line);
            line.resize(row_bytes);
        }
    }
    return Status::Ok();
}

inline Status load_png(const std::string &path, Image &out)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) return Status::Err("cannot open '" + path + "'");
    std::vector<uint8_t> bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    return decode_png(bytes.data(), bytes.size(), out);
}

// image::open(path).to_rgba() (src/renderer/types.rs:161-170): the decoder is chosen by what the
// file holds, not by its name.
inline Status load_image(const std::string &path, Image &out)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) return Status::Err("cannot open '" + path + "'");
    std::vector<uint8_t> bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (bytes.size() >= 8 && bytes[0] == 0x89 && bytes[1] == 'P' && bytes[2] == 'N' && bytes[3] == 'G')
        return decode_png(bytes.data(), bytes.size(), out);
    if (bytes.size() >= 2 && bytes[0] == 0xFF && bytes[1] == 0xD8) return decode_jpeg(bytes.data(), bytes.size(), out);
    return Status::Err("'" + path + "' is neither a JPEG nor a PNG file");
}

} // namespace sxe_host

#endif
