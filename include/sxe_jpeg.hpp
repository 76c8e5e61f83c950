// sxe_jpeg.hpp -- JPEG -> RGBA8 loader for texture files (C++17, header-only, host only).
//
// The reference reads its textures with `image::open(path)?.to_rgba()` and uploads the bytes as
// Rgba8Srgb (src/renderer/types.rs:158-188); its three texture files are JPEGs
// (src/main.rs:152-157).  This is the stand-in for that call on the C++ host side: it decodes
// Huffman-coded JPEG, sequential (SOF0/SOF1, in one scan or one per component) and progressive
// (SOF2: spectral selection and successive approximation, T.81 annex G), 8 bit, one or three
// components, sampling factors up to 2x2, restart intervals, into tightly packed RGBA8 with alpha
// 255, ready for HipDrawer::provide_image.  Lossless, hierarchical and arithmetic-coded files are
// refused with a message.
// Decoded values follow the JFIF definitions (floating-point inverse DCT, BT.601 full-range
// YCbCr, chroma replicated, not interpolated); other decoders differ from it by their integer
// approximations, a few code values at most.  The reference's own texture files are not
// available (LFS pointers), so there is no reference output to compare with: parity unpinned.
#ifndef SXE_JPEG_HPP
#define SXE_JPEG_HPP

#include "sxe_drawer.hpp"

#include <cmath>
#include <cstring>
#include <fstream>

namespace sxe_host {
namespace jpeg_detail {

struct Huffman {
    // canonical code tables as in ITU T.81 annex F.2.2.3
    int mincode[17] = {}, maxcode[18] = {}, valptr[17] = {};
    uint8_t values[256] = {};
    bool present = false;
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int blocks_w = 0, blocks_h = 0; // allocated size in 8x8 blocks (whole MCUs)
    int pred = 0;
    std::vector<uint8_t> plane;     // blocks_w * 8 by blocks_h * 8 samples
};

struct BitReader {
    const uint8_t *p = nullptr, *end = nullptr;
    uint32_t acc = 0;
    int n = 0;
    bool hit_marker = false;
    int bit()
    {
        if (n == 0) {
            uint8_t b = 0;
            if (p < end && !hit_marker) {
                b = *p++;
                if (b == 0xFF) {
                    if (p < end && *p == 0x00) ++p;       // stuffed zero
                    else { hit_marker = true; --p; b = 0; } // a marker: feed zeros from here on
                }
            }
            acc = b;
            n = 8;
        }
        --n;
        return (int)((acc >> n) & 1u);
    }
    int bits(int count)
    {
        int v = 0;
        for (int i = 0; i < count; ++i) v = (v << 1) | bit();
        return v;
    }
    void reset() { n = 0; acc = 0; hit_marker = false; }
};

inline int extend(int v, int t) { return t == 0 ? 0 : (v < (1 << (t - 1)) ? v - (1 << t) + 1 : v); }

inline bool decode_symbol(BitReader &br, const Huffman &h, int &sym)
{
    int code = 0;
    for (int len = 1; len <= 16; ++len) {
        code = (code << 1) | br.bit();
        if (h.maxcode[len] >= 0 && code <= h.maxcode[len] && code >= h.mincode[len]) {
            sym = h.values[h.valptr[len] + code - h.mincode[len]];
            return true;
        }
    }
    return false;
}

static const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                    41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                    30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// Separable inverse DCT straight from the definition (T.81 A.3.3), level shift, round, clamp.
inline void idct_block(const float in[64], uint8_t *dst, int stride)
{
    static float c[8][8];
    static bool ready = false;
    if (!ready) {
        for (int x = 0; x < 8; ++x)
            for (int u = 0; u < 8; ++u)
                c[x][u] = (u == 0 ? std::sqrt(0.125f) : 0.5f) * std::cos((2 * x + 1) * u * 3.14159265358979323846f / 16.0f);
        ready = true;
    }
    float tmp[64];
    for (int y = 0; y < 8; ++y)      // rows: over u
        for (int x = 0; x < 8; ++x) {
            float s = 0.0f;
            for (int u = 0; u < 8; ++u) s += c[x][u] * in[8 * y + u];
            tmp[8 * y + x] = s;
        }
    for (int x = 0; x < 8; ++x)      // columns: over v
        for (int y = 0; y < 8; ++y) {
            float s = 0.0f;
            for (int v = 0; v < 8; ++v) s += c[y][v] * tmp[8 * v + x];
            const float r = std::floor(s + 128.5f);
            dst[y * stride + x] = (uint8_t)(r < 0.0f ? 0.0f : (r > 255.0f ? 255.0f : r));
        }
}

inline uint8_t clamp8(float v)
{
    const float r = std::floor(v + 0.5f);
    return (uint8_t)(r < 0.0f ? 0.0f : (r > 255.0f ? 255.0f : r));
}

// One scan of a frame: which components it carries and which part of the coefficients
// (T.81 B.2.3: spectral selection Ss..Se, successive approximation Ah, Al; a sequential scan has
// 0..63, 0, 0).
struct Scan {
    std::vector<Component *> comps;
    int ss = 0, se = 63, ah = 0, al = 0;
};

// The coefficients of one 8x8 block from a scan, into `coef` (natural order, not yet dequantised).
// Sequential: the whole block (T.81 F.2.2).  Progressive (T.81 G.1.2): the DC coefficient or a band
// of AC coefficients, first pass or one more bit of precision; `eobrun` counts the blocks whose band
// has ended, across calls.
inline const char *decode_block(BitReader &br, bool progressive, const Scan &sc, Component &c, const Huffman *dc, const Huffman *ac,
                                int16_t *coef, int &eobrun)
{
    if (!progressive) {
        int t = 0;
        if (!decode_symbol(br, dc[c.td], t) || t > 11) return "bad DC code";
        c.pred += extend(br.bits(t), t);
        // (a DC value of an 8-bit image fits 12 bits; anything beyond is a damaged or crafted
        // file, and unchecked sums would overflow)
        if (c.pred < -32768 || c.pred > 32767) return "DC value out of range";
        coef[0] = (int16_t)c.pred;
        for (int k = 1; k < 64;) {
            int rs = 0;
            if (!decode_symbol(br, ac[c.ta], rs)) return "bad AC code";
            const int r = rs >> 4, s = rs & 15;
            if (s == 0) {
                if (r == 15) { k += 16; continue; }
                break; // end of block
            }
            k += r;
            if (k > 63) return "AC run past the end of a block";
            coef[kZigzag[k]] = (int16_t)extend(br.bits(s), s);
            ++k;
        }
        return nullptr;
    }
    if (sc.ss == 0) { // DC: the prediction runs over the values shifted down by Al
        if (sc.ah == 0) {
            int t = 0;
            if (!decode_symbol(br, dc[c.td], t) || t > 11) return "bad DC code";
            c.pred += extend(br.bits(t), t);
            if (c.pred < -32768 || c.pred > 32767) return "DC value out of range";
            coef[0] = (int16_t)(c.pred * (1 << sc.al));
        } else if (br.bit()) {
            coef[0] = (int16_t)(coef[0] | (1 << sc.al));
        }
        return nullptr;
    }
    if (sc.ah == 0) { // AC band, first pass
        if (eobrun > 0) { --eobrun; return nullptr; }
        for (int k = sc.ss; k <= sc.se; ++k) {
            int rs = 0;
            if (!decode_symbol(br, ac[c.ta], rs)) return "bad AC code";
            const int r = rs >> 4, s = rs & 15;
            if (s) {
                k += r;
                if (k > sc.se) return "AC run past the end of a band";
                coef[kZigzag[k]] = (int16_t)(extend(br.bits(s), s) * (1 << sc.al));
            } else if (r == 15) {
                k += 15; // sixteen zeros (the loop adds the last step)
            } else {
                eobrun = (1 << r) - 1 + (r ? br.bits(r) : 0); // this block and eobrun more end here
                break;
            }
        }
        return nullptr;
    }
    // AC band, one more bit (T.81 G.1.2.3): coefficients that are already non-zero get a correction
    // bit each as they are passed; a newly non-zero one is +-1 << Al after `r` still-zero ones
    const int p1 = 1 << sc.al, m1 = -(1 << sc.al);
    auto refine = [&](int16_t &v) {
        if (br.bit() && (v & p1) == 0) v = (int16_t)(v + (v >= 0 ? p1 : m1));
    };
    int k = sc.ss;
    if (eobrun == 0) {
        for (; k <= sc.se; ++k) {
            int rs = 0;
            if (!decode_symbol(br, ac[c.ta], rs)) return "bad AC code";
            int r = rs >> 4, s = rs & 15;
            if (s) {
                if (s != 1) return "bad refinement code";
                s = br.bit() ? p1 : m1;
            } else if (r != 15) {
                eobrun = (1 << r) + (r ? br.bits(r) : 0); // the band ends in this block: the rest is refined below
                break;
            }
            while (k <= sc.se) {
                int16_t &v = coef[kZigzag[k]];
                if (v != 0) {
                    refine(v);
                } else if (--r < 0) {
                    break;
                }
                ++k;
            }
            if (s) {
                if (k > sc.se) return "AC run past the end of a band";
                coef[kZigzag[k]] = (int16_t)s;
            }
        }
    }
    if (eobrun > 0) {
        for (; k <= sc.se; ++k)
            if (coef[kZigzag[k]] != 0) refine(coef[kZigzag[k]]);
        --eobrun;
    }
    return nullptr;
}

} // namespace jpeg_detail

inline Status decode_jpeg(const uint8_t *data, size_t size, Image &out)
{
    using namespace jpeg_detail;
    auto err = [](const std::string &m) { return Status::Err("jpeg: " + m); };
    if (size < 4 || data[0] != 0xFF || data[1] != 0xD8) return err("not a JPEG file (no SOI marker)");
    uint16_t quant[4][64] = {};
    bool quant_present[4] = {};
    Huffman dc[4], ac[4];
    std::vector<Component> comps;
    std::vector<std::vector<int16_t>> coefs; // per component: blocks_w * blocks_h blocks of 64, natural order
    int width = 0, height = 0, restart_interval = 0, hmax = 1, vmax = 1, mcus_x = 0, mcus_y = 0;
    bool progressive = false, any_scan = false, seen_eoi = false;
    size_t pos = 2;
    auto be16 = [&](size_t at) { return (int)((data[at] << 8) | data[at + 1]); };
    while (!seen_eoi) {
        if (pos + 2 > size) {
            if (any_scan) break; // (a file cut off behind its last scan: what was decoded stands)
            return err("truncated before the scan");
        }
        if (data[pos] != 0xFF) return err("marker expected");
        while (pos < size && data[pos] == 0xFF) ++pos; // fill bytes
        if (pos >= size) return err("truncated");
        const int marker = data[pos++];
        if (marker == 0xD9) {
            if (!any_scan) return err("no image data");
            seen_eoi = true;
            break;
        }
        if (pos + 2 > size) return err("truncated segment");
        const int len = be16(pos);
        if (len < 2 || pos + (size_t)len > size) return err("bad segment length");
        const uint8_t *seg = data + pos + 2;
        const int n = len - 2;
        if (marker == 0xC0 || marker == 0xC1 || marker == 0xC2) {
            if (!comps.empty()) return err("more than one frame header");
            progressive = marker == 0xC2;
            if (n < 6) return err("short frame header");
            if (seg[0] != 8) return err("only 8-bit samples are supported");
            height = (seg[1] << 8) | seg[2];
            width = (seg[3] << 8) | seg[4];
            const int nc = seg[5];
            if (width <= 0 || height <= 0) return err("empty image");
            if ((nc != 1 && nc != 3) || n < 6 + 3 * nc) return err("only grey and three-component images are supported");
            comps.resize((size_t)nc);
            for (int i = 0; i < nc; ++i) {
                comps[(size_t)i].id = seg[6 + 3 * i];
                comps[(size_t)i].h = seg[7 + 3 * i] >> 4;
                comps[(size_t)i].v = seg[7 + 3 * i] & 15;
                comps[(size_t)i].tq = seg[8 + 3 * i];
                if (comps[(size_t)i].h < 1 || comps[(size_t)i].h > 2 || comps[(size_t)i].v < 1 || comps[(size_t)i].v > 2 || comps[(size_t)i].tq > 3)
                    return err("unsupported sampling factors");
                hmax = std::max(hmax, comps[(size_t)i].h);
                vmax = std::max(vmax, comps[(size_t)i].v);
            }
            if (nc == 1) {
                // A scan with one component is not interleaved (T.81 A.2.2): one 8x8 block per MCU
                // in raster order, whatever sampling factors the frame header carries.
                comps[0].h = comps[0].v = 1;
                hmax = vmax = 1;
            }
            const int mcu_w = 8 * hmax, mcu_h = 8 * vmax;
            mcus_x = (width + mcu_w - 1) / mcu_w;
            mcus_y = (height + mcu_h - 1) / mcu_h;
            if ((size_t)mcus_x * mcus_y * (size_t)(mcu_w * mcu_h) > (size_t)1 << 30) return err("image too large");
            coefs.resize(comps.size());
            for (size_t i = 0; i < comps.size(); ++i) {
                Component &c = comps[i];
                c.blocks_w = mcus_x * c.h;
                c.blocks_h = mcus_y * c.v;
                coefs[i].assign((size_t)c.blocks_w * (size_t)c.blocks_h * 64, 0);
            }
        } else if (marker == 0xC3 || (marker >= 0xC5 && marker <= 0xCF && marker != 0xC8 && marker != 0xCC)) {
            return err("lossless, hierarchical and arithmetic-coded files are not supported (Huffman-coded sequential and progressive only)");
        } else if (marker == 0xDB) {
            int at = 0;
            while (at < n) {
                const int pq = seg[at] >> 4, tq = seg[at] & 15;
                ++at;
                if (tq > 3 || at + (pq ? 128 : 64) > n) return err("bad quantisation table");
                for (int i = 0; i < 64; ++i) {
                    quant[tq][kZigzag[i]] = pq ? (uint16_t)((seg[at] << 8) | seg[at + 1]) : seg[at];
                    at += pq ? 2 : 1;
                }
                quant_present[tq] = true;
            }
        } else if (marker == 0xC4) {
            int at = 0;
            while (at < n) {
                if (at + 17 > n) return err("bad Huffman table");
                const int tc = seg[at] >> 4, th = seg[at] & 15;
                if (tc > 1 || th > 3) return err("bad Huffman table id");
                Huffman &h = tc ? ac[th] : dc[th];
                int total = 0, code = 0, k = 0;
                for (int l = 1; l <= 16; ++l) total += seg[at + l];
                if (total > 256 || at + 17 + total > n) return err("bad Huffman table size");
                for (int l = 1; l <= 16; ++l) {
                    const int cnt = seg[at + l];
                    h.valptr[l] = k;
                    h.mincode[l] = code;
                    h.maxcode[l] = cnt ? code + cnt - 1 : -1;
                    code = (code + cnt) << 1;
                    k += cnt;
                }
                std::memcpy(h.values, seg + at + 17, (size_t)total);
                h.present = true;
                at += 17 + total;
            }
        } else if (marker == 0xDD) {
            if (n < 2) return err("bad restart interval");
            restart_interval = (seg[0] << 8) | seg[1];
        } else if (marker == 0xDA) {
            if (comps.empty()) return err("scan before frame header");
            const int ns = n >= 1 ? seg[0] : 0;
            if (ns < 1 || ns > (int)comps.size() || n < 1 + 2 * ns + 3) return err("bad scan header");
            Scan sc;
            for (int i = 0; i < ns; ++i) {
                Component *c = nullptr;
                for (Component &k : comps)
                    if (k.id == seg[1 + 2 * i]) c = &k;
                if (!c) return err("scan names an unknown component");
                c->td = seg[2 + 2 * i] >> 4;
                c->ta = seg[2 + 2 * i] & 15;
                if (c->td > 3 || c->ta > 3 || !quant_present[c->tq]) return err("scan uses a table that was not defined");
                sc.comps.push_back(c);
            }
            sc.ss = seg[1 + 2 * ns];
            sc.se = seg[2 + 2 * ns];
            sc.ah = seg[3 + 2 * ns] >> 4;
            sc.al = seg[3 + 2 * ns] & 15;
            if (!progressive) {
                if (sc.ss != 0 || sc.se != 63 || sc.ah != 0 || sc.al != 0) return err("bad scan parameters for a sequential file");
            } else if (sc.ss > sc.se || sc.se > 63 || sc.al > 13 || sc.ah > 13 || (sc.ss == 0 && sc.se != 0) || (sc.ss > 0 && ns != 1)) {
                return err("bad scan parameters for a progressive file");
            }
            for (Component *c : sc.comps) {
                const bool need_dc = !progressive || (sc.ss == 0 && sc.ah == 0), need_ac = !progressive || sc.ss > 0;
                if ((need_dc && !dc[c->td].present) || (need_ac && !ac[c->ta].present)) return err("scan uses a table that was not defined");
            }
            // ---- the scan's entropy-coded data ----
            BitReader br;
            br.p = data + pos + (size_t)len;
            br.end = data + size;
            for (Component &c : comps) c.pred = 0;
            int eobrun = 0, until_restart = restart_interval;
            // one component: its own grid of blocks (as many as hold its samples); several:
            // interleaved, MCU by MCU, h x v blocks of each component per MCU (T.81 A.2)
            const bool interleaved = sc.comps.size() > 1;
            Component &c0 = *sc.comps[0];
            const int units_x = interleaved ? mcus_x : ((width * c0.h + hmax - 1) / hmax + 7) / 8;
            const int units_y = interleaved ? mcus_y : ((height * c0.v + vmax - 1) / vmax + 7) / 8;
            for (int uy = 0; uy < units_y; ++uy)
                for (int ux = 0; ux < units_x; ++ux) {
                    if (restart_interval && until_restart == 0) {
                        // the next bytes are an RSTn marker: skip it, reset predictions and the bit buffer
                        const uint8_t *q = br.p;
                        while (q + 1 < br.end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) ++q;
                        if (q + 1 >= br.end) return err("restart marker missing");
                        br.p = q + 2;
                        br.reset();
                        for (Component &c : comps) c.pred = 0;
                        eobrun = 0;
                        until_restart = restart_interval;
                    }
                    for (Component *cp : sc.comps) {
                        Component &c = *cp;
                        std::vector<int16_t> &store = coefs[(size_t)(cp - comps.data())];
                        const int nbx = interleaved ? c.h : 1, nby = interleaved ? c.v : 1;
                        for (int by = 0; by < nby; ++by)
                            for (int bx = 0; bx < nbx; ++bx) {
                                const int X = interleaved ? ux * c.h + bx : ux, Y = interleaved ? uy * c.v + by : uy;
                                int16_t *coef = store.data() + ((size_t)Y * (size_t)c.blocks_w + (size_t)X) * 64;
                                if (const char *what = decode_block(br, progressive, sc, c, dc, ac, coef, eobrun)) return err(what);
                            }
                    }
                    if (restart_interval) --until_restart;
                }
            any_scan = true;
            // on to the marker behind the scan's data (restart markers and stuffed bytes belong to the data)
            const uint8_t *q = br.p;
            while (q + 1 < br.end && !(q[0] == 0xFF && q[1] != 0x00 && !(q[1] >= 0xD0 && q[1] <= 0xD7) && q[1] != 0xFF)) ++q;
            pos = (size_t)(q - data);
            if (pos + 1 >= size) break; // (no end marker: what was decoded stands)
            continue;
        }
        pos += (size_t)len;
    }
    if (!any_scan) return err("no image data");

    // ---- dequantisation and inverse DCT of every block ----
    for (size_t i = 0; i < comps.size(); ++i) {
        Component &c = comps[i];
        c.plane.assign((size_t)c.blocks_w * 8 * (size_t)c.blocks_h * 8, 0);
        for (int by = 0; by < c.blocks_h; ++by)
            for (int bx = 0; bx < c.blocks_w; ++bx) {
                const int16_t *q = coefs[i].data() + ((size_t)by * (size_t)c.blocks_w + (size_t)bx) * 64;
                float coef[64];
                for (int k = 0; k < 64; ++k) coef[k] = (float)((int)q[k] * (int)quant[c.tq][k]);
                idct_block(coef, c.plane.data() + (size_t)(by * 8) * (size_t)(c.blocks_w * 8) + (size_t)(bx * 8), c.blocks_w * 8);
            }
    }

    // ---- colour conversion (JFIF: BT.601 full range), chroma replicated to full resolution ----
    out.width = (uint32_t)width;
    out.height = (uint32_t)height;
    out.rgba.assign((size_t)width * (size_t)height * 4, 255);
    auto sample = [&](const Component &c, int x, int y) {
        const int sx = x * c.h / hmax, sy = y * c.v / vmax;
        return (float)c.plane[(size_t)sy * (size_t)(c.blocks_w * 8) + (size_t)sx];
    };
    for (int y = 0; y < height; ++y)
        for (int x = 0; x < width; ++x) {
            uint8_t *px = out.rgba.data() + ((size_t)y * (size_t)width + (size_t)x) * 4;
            const float Y = sample(comps[0], x, y);
            if (comps.size() == 1) {
                px[0] = px[1] = px[2] = (uint8_t)Y;
            } else {
                const float cb = sample(comps[1], x, y) - 128.0f, cr = sample(comps[2], x, y) - 128.0f;
                px[0] = clamp8(Y + 1.402f * cr);
                px[1] = clamp8(Y - 0.344136f * cb - 0.714136f * cr);
                px[2] = clamp8(Y + 1.772f * cb);
            }
        }
    return Status::Ok();
}

// image::open(path)?.to_rgba() for a JPEG file (src/renderer/types.rs:164-170).
inline Status load_jpeg(const std::string &path, Image &out)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) return Status::Err("cannot open '" + path + "'");
    std::vector<uint8_t> bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    return decode_jpeg(bytes.data(), bytes.size(), out);
}

} // namespace sxe_host

#endif
