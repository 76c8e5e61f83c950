// sxe_raster.hip -- the tile kernel of the raster/shade path (gfx950).
//
// k_raster_shade: one workgroup per 32x32 screen tile.
//   1. coverage + depth: every bin entry is tested against the tile's pixels with exact fixed-point
//      edge functions and the top-left rule; depth test Less + write (src/renderer/drawer.rs:698-705)
//      is a 64-bit minimum of (depth bits | submission order) per pixel in LDS, so the result does
//      not depend on the order in which entries are processed;
//   2. resolve: the winning triangle of every pixel is shaded once (standard.frag:12: bilinear sRGB
//      texture fetch with perspective-correct uv), blended over the cleared target
//      (BlendState::ALPHA, drawer.rs:660-663), encoded to sRGB8 and stored together with the
//      depth; the clear (drawer.rs:307-308) is fused, so the frame is written exactly once.
//
// Numeric contract: every float expression is evaluated as written (-ffp-contract=off).
#include "sxe_device.h"
#include "sxe_srgb_tables.h"

#include <mutex>

namespace sxe {

__constant__ float c_srgb_decode[256];
__constant__ float c_unorm8[256];

// Linear value -> sRGB8 code = number of encode thresholds <= x.  The floats from 2^-13 (below the
// first threshold) to 1.0 are cut into buckets that share sign, exponent and seven mantissa bits;
// no bucket holds more than one threshold (checked when the table is built), so the code is the
// count of thresholds below the bucket plus one comparison of the low 16 bits.
//   entry = count << 24 | low 16 bits of the bucket's threshold (0x10000: the bucket has none)
// Entry 0 stands for everything below 2^-13 (code 0), the last one for x >= 1 (code 255).
constexpr int kEncKeyLo = (127 - 13) << 7;            // float bits >> 16 of 2^-13
constexpr int kEncKeyHi = 127 << 7;                   // float bits >> 16 of 1.0
constexpr int kEncEntries = kEncKeyHi - kEncKeyLo + 2; // 1666
constexpr int kEncPadded = (kEncEntries + 3) & ~3;

__device__ inline uint32_t srgb_encode_lds(const uint32_t *enc, float x)
{
    // (a NaN encodes as 0, like everything that is not above the first threshold: the larger of a
    // NaN and 0 is 0; without this its sign bit would decide between 0 and 255)
    const int bits = __float_as_int(fmaxf(x, 0.0f));
    const int idx = min(max((bits >> 16) - (kEncKeyLo - 1), 0), kEncEntries - 1);
    const uint32_t e = enc[idx];
    return (e >> 24) + (((uint32_t)bits & 0xFFFFu) >= (e & 0x1FFFFu) ? 1u : 0u);
}

__device__ uint4 g_srgb_encode[kEncPadded / 4];

// Builds the bucket table from the threshold list; fails if a bucket would need two comparisons.
static bool build_encode_table(uint32_t *table)
{
    auto bits_of = [](float f) { uint32_t u; __builtin_memcpy(&u, &f, 4); return u; };
    if ((int)(bits_of(SXE_SRGB_ENCODE_THR[0]) >> 16) < kEncKeyLo || !(SXE_SRGB_ENCODE_THR[254] < 1.0f)) return false;
    int k = 0; // thresholds below the current bucket
    table[0] = 0x10000u;
    for (int i = 1; i <= kEncKeyHi - kEncKeyLo; ++i) {
        const uint32_t key = (uint32_t)(kEncKeyLo + i - 1);
        uint32_t entry = (uint32_t)k << 24 | 0x10000u;
        int inside = 0;
        while (k < 255 && (bits_of(SXE_SRGB_ENCODE_THR[k]) >> 16) == key) {
            entry = (entry & 0xFF000000u) | (bits_of(SXE_SRGB_ENCODE_THR[k]) & 0xFFFFu);
            ++k;
            ++inside;
        }
        if (inside > 1) return false;
        table[i] = entry;
    }
    if (k != 255) return false;
    for (int i = kEncEntries - 1; i < kEncPadded; ++i) table[i] = 255u << 24 | 0x10000u;
    return true;
}

static hipError_t upload_encode_table()
{
    static uint32_t table[kEncPadded];
    if (!build_encode_table(table)) return hipErrorInvalidValue;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_srgb_encode), table, sizeof(table));
}

// The same lookup as srgb_encode_lds, on the host (for tests of the table without a device).
static uint32_t srgb_encode_host(const uint32_t *enc, float x)
{
    int bits;
    if (!(x > 0.0f)) x = 0.0f;
    __builtin_memcpy(&bits, &x, 4);
    int idx = (bits >> 16) - (kEncKeyLo - 1);
    idx = idx < 0 ? 0 : (idx > kEncEntries - 1 ? kEncEntries - 1 : idx);
    const uint32_t e = enc[idx];
    return (e >> 24) + (((uint32_t)bits & 0xFFFFu) >= (e & 0x1FFFFu) ? 1u : 0u);
}

// Uploads the constant tables of the tile kernel to `device` (which must be current).  Called by
// sxe_create, once per device however many threads create contexts at the same time; the launch
// path does not check again.
hipError_t init_device_tables(int device)
{
    static std::mutex lock;
    static bool ready[kMaxDevices] = {};
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> hold(lock);
    if (ready[device]) return hipSuccess;
    hipError_t e;
    if ((e = hipMemcpyToSymbol(HIP_SYMBOL(c_srgb_decode), SXE_SRGB_DECODE, sizeof(float) * 256)) != hipSuccess) return e;
    if ((e = hipMemcpyToSymbol(HIP_SYMBOL(c_unorm8), SXE_UNORM8, sizeof(float) * 256)) != hipSuccess) return e;
    if ((e = upload_encode_table()) != hipSuccess) return e;
    ready[device] = true;
    return hipSuccess;
}

constexpr int kWideQueue = 112;   // wide-box entries of one tile that all four waves share
struct WideEntry {                // a queued triangle: corners, depth plane, reciprocal area, submission order
    int32_t x0, y0, x1, y1, x2, y2;
    float z0, dz1, dz2, rA;
    uint32_t order;
    uint32_t box;                 // its pixel box inside the tile, see pack_box
};
// Boxes of the queue up to this many pixels (narrow triangles only) are walked by groups of 16
// lanes, sixteen triangles at a time per workgroup; larger ones by all 256 threads together.
constexpr uint32_t kGroupBox = 128;
constexpr uint32_t kBoxNarrow = 1u << 20;
__device__ inline uint32_t pack_box(int32_t bx0, int32_t bx1, int32_t by0, int32_t by1, int32_t tpx, int32_t tpy, bool is_narrow)
{
    return (uint32_t)(bx0 - tpx) | ((uint32_t)(by0 - tpy) << 5) | ((uint32_t)(bx1 - bx0) << 10) | ((uint32_t)(by1 - by0) << 15) |
           (is_narrow ? kBoxNarrow : 0u);
}
__device__ inline bool box_for_group(uint32_t box)
{
    return (box & kBoxNarrow) && (((box >> 10) & 31u) + 1u) * (((box >> 15) & 31u) + 1u) <= kGroupBox;
}
constexpr int kTexCache = 64;     // texture descriptors kept in LDS
constexpr int kEntriesPerLane = 2; // bin entries a lane fetches together

__device__ inline uint32_t unorm8_encode(float a)
{
    if (!(a > 0.0f)) a = 0.0f;
    if (a > 1.0f) a = 1.0f;
    return (uint32_t)(int)rintf(a * 255.0f);
}

// texture(sampler2D, uv) for a single-level RGBA8 sRGB image, linear filter, clamp to edge
// (standard.frag:12; sampler at src/renderer/drawer.rs:231; format at src/renderer/types.rs:186),
// split into texel addressing and filtering so that the four fetches of several pixels overlap.
// The textures are kept as tables of 8-byte cells: cell (c, j), c in [-1, width], j in
// [-1, height - 1], holds texels (c, j) and (c, j + 1), with the clamp to the edge already applied.
// The four texels a sample whose upper left texel is (i, j) filters are the cells (i, j) and
// (i + 1, j), neighbours in memory: a pixel's fetch is one 16-byte load (at an 8-byte boundary)
// instead of four loads from two lines.  Twice the texels' own size; a table of whole 16-byte
// footprints, four times their size, has every load aligned and the tile kernel 5 % slower on
// minified textures, whose every pixel is a miss: the smaller table misses the L2 less often.
// Two neighbouring cells of a table: the texels (i, j), (i, j + 1), (i + 1, j), (i + 1, j + 1).
__device__ inline uint4 load_footprint(const uint2 *cells, uint32_t entry)
{
    typedef uint32_t u32x4_a8 __attribute__((ext_vector_type(4), aligned(8)));
    const u32x4_a8 v = *reinterpret_cast<const u32x4_a8 *>(cells + entry);
    return make_uint4(v.x, v.y, v.z, v.w);
}

struct TexelQuad {
    uint32_t entry; // the footprint's first cell in the pooled tables
    float ax, ay;
};

__device__ inline TexelQuad texel_quad(const TexDev &t, float u, float v)
{
    const float fw = (float)t.width, fh = (float)t.height;
    float x = u * fw - 0.5f, y = v * fh - 0.5f;
    if (!(x >= -1.0f)) x = -1.0f;
    if (x > fw) x = fw;
    if (!(y >= -1.0f)) y = -1.0f;
    if (y > fh) y = fh;
    const float fx0 = floorf(x), fy0 = floorf(y);
    TexelQuad q;
    q.ax = x - fx0;
    q.ay = y - fy0;
    // fx0 is in [-1, width]; at width the weight ax is 0 and the footprint at width - 1, whose two
    // columns are both the last texel, gives the same sums.  Likewise for the rows.
    const uint32_t i = (uint32_t)(min((int)fx0, (int)t.width - 1) + 1), j = (uint32_t)(min((int)fy0, (int)t.height - 1) + 1);
    // rows and widths are at most 16384 (checked when the textures are set): the full-rate 24-bit
    // multiply gives the same product as the general one, which costs four times as much
    q.entry = t.offset + __umul24(j, t.width + 2u) + i;
    return q;
}

// sRGB decode of every texel before filtering.  Only the colour channels: the tiles this kernel
// renders hold only triangles whose texture is opaque (TexDev::flags == 0: k_setup flags every tile
// that a triangle with a translucent texture touches with kTileBlend, and those tiles are left to
// k_raster_blend), so every texel's alpha is 255, and filtering four ones gives exactly one.
__device__ inline void filter_quad(const float *dec, uint32_t t00, uint32_t t10, uint32_t t01, uint32_t t11, float ax,
                                   float ay, float out[4])
{
    for (int k = 0; k < 3; ++k) {
        const int sh = 8 * k;
        const float c00 = dec[(t00 >> sh) & 255], c10 = dec[(t10 >> sh) & 255];
        const float c01 = dec[(t01 >> sh) & 255], c11 = dec[(t11 >> sh) & 255];
        const float top = c00 + ax * (c10 - c00);
        const float bot = c01 + ax * (c11 - c01);
        out[k] = top + ay * (bot - top);
    }
    out[3] = 1.0f;
}

// The stored colour in the byte order of the target: R, G, B, A as shaded, or B, G, R, A
// (src/renderer/presenter.rs:472-478 takes whichever sRGB format the surface lists first).
__device__ inline uint32_t target_order(uint32_t rgba, uint32_t bgra)
{
    return bgra ? __builtin_amdgcn_perm(rgba, rgba, 0x03000102u) : rgba;
}

// A finished pixel goes to its images with the streaming hint: nothing on this GPU reads the images
// before the frame is complete, and as ordinary dirty lines 66 MB of them per 4K frame sit in the
// L2 while the records and texture tables of this frame, and the geometry kernel of the next one,
// want it (2 % of the frame).
__device__ inline void store_pixel(const RasterArgs &a, size_t p, uint32_t color, float depth)
{
    __builtin_nontemporal_store(color, reinterpret_cast<uint32_t *>(a.out_color) + p);
    if (a.out_depth) __builtin_nontemporal_store(depth, a.out_depth + p);
}

__device__ inline float clamp_depth(float z)
{
    if (!(z > 0.0f)) z = 0.0f;
    return z;
}

// z is not negative (clamp_depth).  No read first: the LDS read would have to come back before the
// lane can go on, the atomic does not.
__device__ inline void depth_min(unsigned long long *vis, int pix, float z, uint32_t order)
{
    atomicMin(&vis[pix], ((unsigned long long)__float_as_uint(z) << 32) | order);
}

// One lane walks the pixel box of one narrow triangle with 32-bit incremental edge functions, row
// by row; the step to the next pixel is chosen between "one to the right" and "first of the next
// row" by selects, so the loop body has no branch but the coverage test.
__device__ inline void raster_lane(unsigned long long *vis, const Tri &t, const RastRec &r, uint32_t order, int32_t bx0,
                                   int32_t bx1, int32_t by0, int32_t by1, int32_t tpx, int32_t tpy)
{
    // every difference below is < 2^15 and every product < 2^29: exact in 32 bits
    const EdgeStart e = edge_start(t, bx0, by0);
    const int32_t b1 = e.b1, b2 = e.b2;
    const float rA = r.rA;
    const int32_t last = bx1 - bx0; // column index of a row's last pixel
    const int32_t ax0 = -256 * e.ey0, ax1 = -256 * e.ey1, ax2 = -256 * e.ey2; // one pixel to the right
    // from the last pixel of a row to the first of the next: one down, `last` to the left
    const int32_t aw0 = 256 * e.ex0 - mul_narrow(ax0, last), aw1 = 256 * e.ex1 - mul_narrow(ax1, last),
                  aw2 = 256 * e.ex2 - mul_narrow(ax2, last);
    int32_t E0 = e.R0, E1 = e.R1, E2 = e.R2;
    int32_t col = 0;
    int32_t pix = (by0 - tpy) * kTile + (bx0 - tpx);
    const int32_t pix_wrap = kTile - last;
    const int32_t cnt = mul_narrow(last + 1, by1 - by0 + 1);
    for (int32_t k = 0; k < cnt; ++k) {
        if ((E0 | E1 | E2) >= 0) {
            const float l1 = (float)(E1 + b1) * rA, l2 = (float)(E2 + b2) * rA;
            const float z = clamp_depth((r.z0 + l1 * r.dz1) + l2 * r.dz2);
            depth_min(vis, pix, z, order);
        }
        const bool wrap = col == last;
        E0 += wrap ? aw0 : ax0;
        E1 += wrap ? aw1 : ax1;
        E2 += wrap ? aw2 : ax2;
        pix += wrap ? pix_wrap : 1;
        col = wrap ? 0 : col + 1;
    }
}

// Wide boxes: each lane prepares one triangle (edge functions at the box origin, reciprocal area,
// row divider), then the wave takes the triangles one at a time, 64 pixels per step; `first` and
// `stride` let several waves interleave their steps on the same triangle.
struct WideSetup {
    int32_t ex0, ey0, ex1, ey1, ex2, ey2; // edge vectors
    int32_t R0, R1, R2;                   // biased edge functions at the first pixel of the box
    int32_t b1, b2;
    float rA;
    int32_t qx0, qy0;
    uint32_t bw, cnt, magic;              // idx / bw == (idx * magic) >> 16 for idx < 1024, bw <= 32
    bool is_narrow;
    long long W0, W1, W2;                 // the same edge functions in 64 bits (triangles that are not narrow)
    bool misses;                          // an edge has the whole box on its outside: nothing to draw here
};

__device__ inline WideSetup wide_setup(const Tri &t, bool is_narrow, float rA, int32_t bx0, int32_t bx1, int32_t by0, int32_t by1)
{
    WideSetup w;
    w.ex0 = t.x2 - t.x1; w.ey0 = t.y2 - t.y1;
    w.ex1 = t.x0 - t.x2; w.ey1 = t.y0 - t.y2;
    w.ex2 = t.x1 - t.x0; w.ey2 = t.y1 - t.y0;
    w.b1 = edge_bias(w.ex1, w.ey1);
    w.b2 = edge_bias(w.ex2, w.ey2);
    w.is_narrow = is_narrow;
    const int32_t sx = 256 * bx0 + 128, sy = 256 * by0 + 128;
    // meaningful for narrow triangles only (every product below 2^28); others use the 64-bit loop
    w.R0 = mul_narrow(w.ex0, sy - t.y1) - mul_narrow(w.ey0, sx - t.x1) - edge_bias(w.ex0, w.ey0);
    w.R1 = mul_narrow(w.ex1, sy - t.y2) - mul_narrow(w.ey1, sx - t.x2) - w.b1;
    w.R2 = mul_narrow(w.ex2, sy - t.y0) - mul_narrow(w.ey2, sx - t.x0) - w.b2;
    w.rA = rA;
    w.qx0 = bx0; w.qy0 = by0;
    w.bw = (uint32_t)(bx1 - bx0 + 1);
    w.cnt = w.bw * (uint32_t)(by1 - by0 + 1);
    w.magic = (65536u + w.bw - 1u) / w.bw;
    w.W0 = (long long)w.ex0 * (long long)(sy - t.y1) - (long long)w.ey0 * (long long)(sx - t.x1) - edge_bias(w.ex0, w.ey0);
    w.W1 = (long long)w.ex1 * (long long)(sy - t.y2) - (long long)w.ey1 * (long long)(sx - t.x2) - w.b1;
    w.W2 = (long long)w.ex2 * (long long)(sy - t.y0) - (long long)w.ey2 * (long long)(sx - t.x0) - w.b2;
    // The box is the triangle's bounding box cut to the tile, and a large triangle leaves much of
    // that empty.  An edge function is linear, so its largest value over the box is at a corner.
    const long long sx_span = 256ll * (long long)(bx1 - bx0), sy_span = 256ll * (long long)(by1 - by0);
    const long long m0 = w.W0 + max(0ll, -(long long)w.ey0 * sx_span) + max(0ll, (long long)w.ex0 * sy_span);
    const long long m1 = w.W1 + max(0ll, -(long long)w.ey1 * sx_span) + max(0ll, (long long)w.ex1 * sy_span);
    const long long m2 = w.W2 + max(0ll, -(long long)w.ey2 * sx_span) + max(0ll, (long long)w.ex2 * sy_span);
    w.misses = (m0 | m1 | m2) < 0;
    return w;
}

#define SXE_RL(x) __builtin_amdgcn_readlane((int)(x), src)
#define SXE_RLF(x) __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(x), src))

// Rasterise the triangle prepared by lane `src` with the whole wave.
__device__ inline void raster_wide(unsigned long long *vis, int src, const WideSetup &w, const Tri &t, const RastRec &r,
                                   uint32_t lane_order, uint32_t first, uint32_t stride, int32_t tpx, int32_t tpy)
{
    const float z0 = SXE_RLF(r.z0), dz1 = SXE_RLF(r.dz1), dz2 = SXE_RLF(r.dz2), rA = SXE_RLF(w.rA);
    const uint32_t order = (uint32_t)SXE_RL(lane_order);
    const int32_t qx0 = SXE_RL(w.qx0), qy0 = SXE_RL(w.qy0);
    const uint32_t bw = (uint32_t)SXE_RL(w.bw), cnt = (uint32_t)SXE_RL(w.cnt), magic = (uint32_t)SXE_RL(w.magic);
    if (SXE_RL(w.is_narrow ? 1 : 0)) {
        const int32_t R0 = SXE_RL(w.R0), R1 = SXE_RL(w.R1), R2 = SXE_RL(w.R2), b1 = SXE_RL(w.b1), b2 = SXE_RL(w.b2);
        // one pixel to the right / one pixel down
        const int32_t ax0 = -256 * SXE_RL(w.ey0), ax1 = -256 * SXE_RL(w.ey1), ax2 = -256 * SXE_RL(w.ey2);
        const int32_t ay0 = 256 * SXE_RL(w.ex0), ay1 = 256 * SXE_RL(w.ex1), ay2 = 256 * SXE_RL(w.ex2);
        for (uint32_t idx = first; idx < cnt; idx += stride) {
            const uint32_t row = __umul24(idx, magic) >> 16;
            const int32_t di = (int32_t)(idx - __umul24(row, bw)), dj = (int32_t)row;
            // steps are below 2^22 and pixel offsets below 2^6: 24-bit multiplies are exact
            const int32_t E0 = R0 + mul_narrow(ax0, di) + mul_narrow(ay0, dj);
            const int32_t E1 = R1 + mul_narrow(ax1, di) + mul_narrow(ay1, dj);
            const int32_t E2 = R2 + mul_narrow(ax2, di) + mul_narrow(ay2, dj);
            if ((E0 | E1 | E2) >= 0) {
                const float l1 = (float)(E1 + b1) * rA, l2 = (float)(E2 + b2) * rA;
                const float z = clamp_depth((z0 + l1 * dz1) + l2 * dz2);
                depth_min(vis, (qy0 + dj - tpy) * kTile + (qx0 + di - tpx), z, order);
            }
        }
    } else {
        // 64-bit edge functions at the box origin plus 32 x 32 -> 64 bit steps from there
        const long long W0 = ((long long)SXE_RL((int)(w.W0 >> 32)) << 32) | (unsigned int)SXE_RL((int)(unsigned int)w.W0);
        const long long W1 = ((long long)SXE_RL((int)(w.W1 >> 32)) << 32) | (unsigned int)SXE_RL((int)(unsigned int)w.W1);
        const long long W2 = ((long long)SXE_RL((int)(w.W2 >> 32)) << 32) | (unsigned int)SXE_RL((int)(unsigned int)w.W2);
        const int32_t ex0 = SXE_RL(w.ex0), ey0 = SXE_RL(w.ey0), ex1 = SXE_RL(w.ex1), ey1 = SXE_RL(w.ey1), ex2 = SXE_RL(w.ex2), ey2 = SXE_RL(w.ey2);
        const long long b1 = SXE_RL(w.b1), b2 = SXE_RL(w.b2);
        for (uint32_t idx = first; idx < cnt; idx += stride) {
            const uint32_t row = __umul24(idx, magic) >> 16;
            const int32_t di = (int32_t)(idx - __umul24(row, bw)), dj = (int32_t)row;
            const int32_t sx = 256 * di, sy = 256 * dj;
            const long long E0 = W0 + (long long)ex0 * (long long)sy - (long long)ey0 * (long long)sx;
            const long long E1 = W1 + (long long)ex1 * (long long)sy - (long long)ey1 * (long long)sx;
            const long long E2 = W2 + (long long)ex2 * (long long)sy - (long long)ey2 * (long long)sx;
            if ((E0 | E1 | E2) >= 0) {
                const float l1 = (float)(E1 + b1) * rA, l2 = (float)(E2 + b2) * rA;
                const float z = clamp_depth((z0 + l1 * dz1) + l2 * dz2);
                depth_min(vis, (qy0 + dj - tpy) * kTile + (qx0 + di - tpx), z, order);
            }
        }
    }
}

// What the resolve loop holds about one covered pixel while its records are on their way.
struct PixelWork {
    uint32_t pix, slot; // tile-local pixel index; the order slot of the winning triangle
    float depth;
    ShadeRec h;
    uint4 geo;          // x0, y0 and the packed corner offsets of that triangle (first half of its RastRec)
};

__device__ inline void request_pixel(const RasterArgs &a, const unsigned long long *vis, const uint16_t *s_cov, uint32_t q,
                                     PixelWork &w)
{
    w.pix = s_cov[q];
    const unsigned long long key = vis[w.pix];
    w.slot = (uint32_t)key;
    w.depth = __uint_as_float((uint32_t)(key >> 32));
    w.h = a.shade[w.slot];
    w.geo = reinterpret_cast<const uint4 *>(a.rast)[2 * (size_t)w.slot];
}

// Shading of one covered pixel, in two steps so that the loop can ask for the next pixel's records
// between them, when this pixel's records are no longer needed and its texels are on their way.
struct PixelMid { // what the second step needs
    uint32_t pix;
    int32_t px, py;
    float depth, l1, l2, riw, ax, ay;
    uint32_t prim, slot;
    uint32_t t00, t10, t01, t11;
};

// Step 1: barycentrics of the sample in the winning triangle, perspective division, texel
// addresses; issues the texel loads.
// MANYTEX: the texture table is longer than the kTexCache descriptors kept in LDS.
template <bool MANYTEX>
__device__ inline PixelMid shade_begin(const RasterArgs &a, const PixelWork &w, int32_t tpx, int32_t tpy, const TexDev *s_tex)
{
    PixelMid m;
    const int32_t px = tpx + (int32_t)(w.pix & (kTile - 1)), py = tpy + (int32_t)(w.pix / kTile);
    uint32_t slot = w.slot;
    ShadeRec h = w.h;
    RastRec r;
    r.x0 = (int32_t)w.geo.x; r.y0 = (int32_t)w.geo.y; r.d1 = w.geo.z; r.d2 = w.geo.w;
    r.z0 = r.dz1 = r.dz2 = r.rA = 0.0f; // (the depth plane is not needed any more; the area comes with h)
    if (h.tex_fan >> 24) {
        // A clipped triangle won: which triangle of its fan?  As a rule one of them covers the
        // sample, but the fan's corners are snapped to the grid one by one, and two of its
        // triangles can overlap by a sliver there.  Drawn one after the other under
        // Comparison::Less, the first of those with the smallest depth is the one that stays, and
        // that depth is the pixel's (the raster pass took the minimum over all of them): the first
        // covering triangle whose depth plane gives exactly the winning depth is taken.
        const long long sx = 256 * px + 128, sy = 256 * py + 128;
        const uint32_t fan_base = h.fan_or_ra, n_fan = h.tex_fan >> 24;
        // (should no fan triangle reproduce the winning depth bit for bit, the first one that covers
        // the sample is taken, rather than the primitive's own slot, which holds no planes; the
        // match relies on raster_lane, raster_wide and this loop evaluating the depth with one and
        // the same expression)
        bool found = false;
        for (uint32_t f = 0; f < n_fan; ++f) {
            // (the record's halves one after the other, the depth plane only for a triangle that
            // covers: the whole record held through the loop costs the shading of every pixel registers)
            const uint4 fg = reinterpret_cast<const uint4 *>(a.rast)[2 * (size_t)(fan_base + f)];
            if ((int32_t)fg.x == kNoRaster) continue;
            RastRec fr;
            fr.x0 = (int32_t)fg.x; fr.y0 = (int32_t)fg.y; fr.d1 = fg.z; fr.d2 = fg.w;
            fr.z0 = fr.dz1 = fr.dz2 = fr.rA = 0.0f;
            const Tri ft = unpack_tri(fr, a.wide, fan_base + f);
            const long long fx0 = ft.x2 - ft.x1, fy0 = ft.y2 - ft.y1, fx1 = ft.x0 - ft.x2, fy1 = ft.y0 - ft.y2,
                            fx2 = ft.x1 - ft.x0, fy2 = ft.y1 - ft.y0;
            const long long R1 = fx1 * (sy - ft.y2) - fy1 * (sx - ft.x2), R2 = fx2 * (sy - ft.y0) - fy2 * (sx - ft.x0);
            const long long E0 = fx0 * (sy - ft.y1) - fy0 * (sx - ft.x1) - edge_bias((int32_t)fx0, (int32_t)fy0);
            const long long E1 = R1 - edge_bias((int32_t)fx1, (int32_t)fy1);
            const long long E2 = R2 - edge_bias((int32_t)fx2, (int32_t)fy2);
            if ((E0 | E1 | E2) < 0) continue;
            // (the depth the raster pass gave this triangle's fragment: raster_lane, raster_wide)
            const uint4 fp = reinterpret_cast<const uint4 *>(a.rast)[2 * (size_t)(fan_base + f) + 1]; // z0, dz1, dz2, rA
            const float fl1 = (float)R1 * __uint_as_float(fp.w), fl2 = (float)R2 * __uint_as_float(fp.w);
            const float fz = clamp_depth((__uint_as_float(fp.x) + fl1 * __uint_as_float(fp.y)) + fl2 * __uint_as_float(fp.z));
            const bool exact = __float_as_uint(fz) == __float_as_uint(w.depth);
            if (exact || !found) {
                slot = fan_base + f;
                r = fr;
                found = true;
            }
            if (exact) break;
        }
        h = a.shade[slot];
    }
    const Tri t = unpack_tri(r, a.wide, slot);
    const float rA = __uint_as_float(h.fan_or_ra); // (of the fan triangle that was found, for a clipped one)
    float l1, l2;
    if (r.d1 != kWideTri) { // narrow: same integers as below, from full-rate 24-bit multiplies
        // (the edge functions as edge_start() writes them: the sample and the corners relative to corner 0)
        const int32_t X = 256 * px + 128 - t.x0, Y = 256 * py + 128 - t.y0;
        const int32_t E1 = mul_narrow(t.y2 - t.y0, X) - mul_narrow(t.x2 - t.x0, Y);
        const int32_t E2 = mul_narrow(t.x1 - t.x0, Y) - mul_narrow(t.y1 - t.y0, X);
        l1 = (float)E1 * rA;
        l2 = (float)E2 * rA;
    } else {
        const long long sx = 256 * px + 128, sy = 256 * py + 128;
        const long long E1 = (long long)(t.x0 - t.x2) * (sy - t.y2) - (long long)(t.y0 - t.y2) * (sx - t.x2);
        const long long E2 = (long long)(t.x1 - t.x0) * (sy - t.y0) - (long long)(t.y1 - t.y0) * (sx - t.x0);
        l1 = (float)E1 * rA;
        l2 = (float)E2 * rA;
    }
    const float iw = (h.iw0 + l1 * h.diw1) + l2 * h.diw2;
    const float riw = 1.0f / iw;
    m.pix = w.pix; m.px = px; m.py = py; m.depth = w.depth; m.l1 = l1; m.l2 = l2; m.riw = riw; m.prim = h.prim; m.slot = slot;
    m.ax = m.ay = 0.0f;
    m.t00 = m.t10 = m.t01 = m.t11 = 0u;
    if (a.shading != SXE_SHADE_VERTEX_COLOR) {
        const float u = ((h.uw0 + l1 * h.duw1) + l2 * h.duw2) * riw;
        const float v = ((h.vw0 + l1 * h.dvw1) + l2 * h.dvw2) * riw;
        const uint32_t ti = h.tex_fan & 0xFFFFFFu;
        // (never a choice between the two memories at run time: that would be a load through a
        // generic pointer, which waits on both)
        const TexDev td = MANYTEX ? a.textures[ti] : s_tex[ti];
        const TexelQuad tq = texel_quad(td, u, v);
        m.ax = tq.ax; m.ay = tq.ay;
        const uint4 fp = load_footprint(a.footprints, tq.entry);
        m.t00 = fp.x; m.t10 = fp.z; m.t01 = fp.y; m.t11 = fp.w;
    }
    return m;
}

// Step 2: filter and encode; the pixel's colour replaces its triangle in the visibility buffer
// (the tile's last pass stores whole rows from there); COVERAGE: the primitive id goes to memory.
template <bool COVERAGE>
__device__ inline void shade_end(const RasterArgs &a, const PixelMid &m, const float *s_dec, const uint32_t *s_enc,
                                 unsigned long long *vis)
{
    float src[4];
    if (a.shading == SXE_SHADE_VERTEX_COLOR) {
        const ColorRec c = a.color[m.slot];
        src[0] = ((c.rw0 + m.l1 * c.drw1) + m.l2 * c.drw2) * m.riw;
        src[1] = ((c.gw0 + m.l1 * c.dgw1) + m.l2 * c.dgw2) * m.riw;
        src[2] = ((c.bw0 + m.l1 * c.dbw1) + m.l2 * c.dbw2) * m.riw;
        src[3] = 1.0f;
    } else {
        filter_quad(s_dec, m.t00, m.t10, m.t01, m.t11, m.ax, m.ay, src);
        if (a.shading == SXE_SHADE_COLOR_TIMES_TEXTURE) {
            const ColorRec c = a.color[m.slot];
            src[0] = (((c.rw0 + m.l1 * c.drw1) + m.l2 * c.drw2) * m.riw) * src[0];
            src[1] = (((c.gw0 + m.l1 * c.dgw1) + m.l2 * c.dgw2) * m.riw) * src[1];
            src[2] = (((c.bw0 + m.l1 * c.dbw1) + m.l2 * c.dbw2) * m.riw) * src[2];
        } else if (a.shading == SXE_SHADE_LIT) {
            const float4 lit = reinterpret_cast<const float4 *>(&a.color[m.slot])[2]; // (dbw2, lit_a, lit_s, -)
            src[0] = src[0] * lit.y + lit.z;
            src[1] = src[1] * lit.y + lit.z;
            src[2] = src[2] * lit.y + lit.z;
        }
    }
    // BlendState::ALPHA (src/renderer/drawer.rs:660-663) over the cleared sRGB8 target:
    // C = Cs * As + Cd * (1 - As), A = As + Ad * (1 - As).  As is exactly 1 for every fragment this
    // kernel shades (the textures of a tile that reaches it are opaque, see filter_quad; vertex
    // colours have no alpha), and then the blend returns Cs and 1 bit for bit, so it is not
    // evaluated and only the depth winner is shaded.
    const uint32_t cr = srgb_encode_lds(s_enc, src[0]);
    const uint32_t cg = srgb_encode_lds(s_enc, src[1]);
    const uint32_t cb = srgb_encode_lds(s_enc, src[2]);
    vis[m.pix] = ((unsigned long long)__float_as_uint(m.depth) << 32) | (cr | (cg << 8) | (cb << 16) | 0xFF000000u);
    if (COVERAGE) a.out_prim[(size_t)(__umul24((uint32_t)m.py, a.g.width) + (uint32_t)m.px)] = m.prim; // (frames are at most 8192 wide and high)
}

// (seven resident workgroups per compute unit: at most 72 registers and 22 KB of LDS, which is
// handed out in 1 KB pieces; eight, with a shorter queue of large boxes and 64 registers, gain 1 %)
// COVERAGE: the frame keeps a primitive-id image.  DEV: built with the development switches of
// RasterArgs::debug and the per-tile clock stamps; the ordinary build has none of their tests.
template <bool COVERAGE, bool DEV, bool MANYTEX>
__global__ __launch_bounds__(kRasterThreads) __attribute__((amdgpu_waves_per_eu(7, 7))) void k_raster_shade(RasterArgs a)
{
    if (!DEV) {
        a.debug = 0u;
        a.tile_times = nullptr;
    }
    __shared__ unsigned long long vis[kTilePixels]; // (depth bits << 32) | order, per pixel
    __shared__ __attribute__((aligned(16))) uint32_t s_enc[kEncPadded];
    __shared__ float s_dec[256];
    __shared__ TexDev s_tex[kTexCache];
    // the queue of large boxes and the list of covered pixels are never in use at the same time
    __shared__ union {
        WideEntry wq[kWideQueue];
        struct { uint16_t cov[kTilePixels]; uint32_t bucket[256]; };
    } s_overlay;
    WideEntry *s_wq = s_overlay.wq;
    uint16_t *s_cov = s_overlay.cov; // tile-local indices of the covered pixels
    __shared__ uint32_t s_nwq, s_ncov;
    __shared__ uint32_t s_magic[33]; // idx / w == (idx * s_magic[w]) >> 16 for idx < 1024, w <= 32
    // the frame's counters are final by now: report them to the host without a copy command
    if (blockIdx.x == 0 && a.host_counters && threadIdx.x < sizeof(FrameCounters) / 4)
        reinterpret_cast<uint32_t *>(a.host_counters)[threadIdx.x] = reinterpret_cast<const uint32_t *>(a.counters)[threadIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // What the workgroup reads before it can start is requested in one go -- the overflow flag, the
    // launch record, its share of the tables -- and the flag is acted on behind the first barrier,
    // when all of it has arrived: one round trip to memory instead of two.
    const uint32_t overflow = a.counters->overflow;
    const uint4 launch = reinterpret_cast<const uint4 *>(a.tile_launch)[2 * blockIdx.x];     // tile; entries: late, class 3, 2
    const uint4 launch2 = reinterpret_cast<const uint4 *>(a.tile_launch)[2 * blockIdx.x + 1]; // class 1, 0
    static_assert(kEncPadded / 4 <= 2 * kRasterThreads, "two table pieces per thread");
    const uint4 enc_a = g_srgb_encode[tid], enc_b = g_srgb_encode[min(tid + kRasterThreads, kEncPadded / 4 - 1)];
    const float dec = c_srgb_decode[tid];
    // (a frame that overflowed has no launch records: tile 0 stands in until the workgroup leaves)
    const uint32_t tile = overflow ? 0u : (uint32_t)__builtin_amdgcn_readfirstlane((int)launch.x);
    const uint32_t view = overflow ? 0u : (uint32_t)__builtin_amdgcn_readfirstlane((int)launch2.z);
    // a tile with translucent triangles is rendered by k_raster_blend, in submission order
    if (a.blend_scene && !overflow && ((uint32_t)__builtin_amdgcn_readfirstlane((int)launch2.w) & kTileBlend)) return;
    // every view renders into its own image, one behind the other
    a.out_color += 4 * a.image_stride * view;
    if (a.out_depth) a.out_depth += a.image_stride * view; // (null: the depth images are not kept)
    if (COVERAGE) a.out_prim += a.image_stride * view;
#define SXE_STAMP(i) do { if (a.tile_times && tid == 0) a.tile_times[8 * tile + (i)] = wall_clock64(); } while (0)
    SXE_STAMP(0);
    const int32_t tpx = (int32_t)(tile % a.g.tiles_x) * kTile;
    const int32_t tpy = (int32_t)(tile / a.g.tiles_x + a.g.tile_row0) * kTile;
    for (int i = tid; i < kTilePixels; i += kRasterThreads) vis[i] = kVisEmpty;
    reinterpret_cast<uint4 *>(s_enc)[tid] = enc_a;
    if (tid + kRasterThreads < kEncPadded / 4) reinterpret_cast<uint4 *>(s_enc)[tid + kRasterThreads] = enc_b;
    s_dec[tid] = dec;
    if (tid < kTexCache && (uint32_t)tid < a.n_textures) s_tex[tid] = a.textures[tid];
    if (tid == 0) { s_nwq = 0; s_ncov = 0; }
    if (tid >= 1 && tid <= 32) s_magic[tid] = (65536u + (uint32_t)tid - 1u) / (uint32_t)tid;
    __syncthreads();
    SXE_STAMP(1);
    if (overflow) return;

    // rectangle of this tile that exists in the frame and the scissor (inclusive)
    const int32_t cx0 = tpx, cx1 = min(tpx + kTile - 1, (int32_t)a.g.width - 1);
    const int32_t cy0 = max(tpy, (int32_t)a.g.sy0), cy1 = min(tpy + kTile - 1, (int32_t)a.g.sy1 - 1);

    // The tile's entries are taken region by region, largest boxes first (late entries, then size
    // class 3 down to 0): the 64 consecutive entries a wave takes are of similar size, and the quick
    // ones fill the gaps at the end.  Entry e of that sequence lies in the region whose running
    // total first exceeds e.
    const uint32_t e1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)launch.y);        // late
    const uint32_t e2 = e1 + (uint32_t)__builtin_amdgcn_readfirstlane((int)launch.z);   // + class 3
    const uint32_t e3 = e2 + (uint32_t)__builtin_amdgcn_readfirstlane((int)launch.w);   // + class 2
    const uint32_t e4 = e3 + (uint32_t)__builtin_amdgcn_readfirstlane((int)launch2.x);  // + class 1
    const uint32_t e5 = e4 + (uint32_t)__builtin_amdgcn_readfirstlane((int)launch2.y);  // + class 0
    const uint32_t n = (a.debug & kDebugSkipRaster) ? 0u : e5;
    if (e5 == 0u && !DEV) {
        // nothing was binned to this tile: the clear values, row by row, and that is all
        for (int k = 0; k < kTilePixels / kRasterThreads; ++k) {
            const int32_t i = tpx + (tid & (kTile - 1)), j = tpy + (tid / kTile) + k * (kRasterThreads / kTile);
            if (i > cx1 || j < cy0 || j > cy1) continue;
            const size_t p = (size_t)(__umul24((uint32_t)j, a.g.width) + (uint32_t)i); // (frames are at most 8192 wide and high)
            store_pixel(a, p, target_order(a.clear_code, a.bgra), 1.0f);
            if (COVERAGE) a.out_prim[p] = 0xFFFFFFFFu;
        }
        return;
    }
    const uint32_t *bin = a.bins + (size_t)(view * a.g.n_tiles + tile) * a.bl.stride;
    // start of each region minus the entries taken before it
    const uint32_t o4 = a.bl.off[4], o3 = a.bl.off[3] - e1, o2 = a.bl.off[2] - e2, o1 = a.bl.off[1] - e3, o0 = a.bl.off[0] - e4;
    {
        const uint32_t lo = 0u, hi = n;
        // a wave takes kEntriesPerLane * 64 consecutive entries per round: all their slot numbers
        // are loaded together, then all their records, so a round costs two memory round trips
        for (uint32_t base = lo; base < hi; base += kRasterThreads * kEntriesPerLane) {
          uint32_t slots[kEntriesPerLane];
          RastRec recs[kEntriesPerLane];
#pragma unroll
          for (int k = 0; k < kEntriesPerLane; ++k) {
              const uint32_t e = base + (uint32_t)(kRasterThreads * k + tid);
              const uint32_t at = e + (e < e1 ? o4 : e < e2 ? o3 : e < e3 ? o2 : e < e4 ? o1 : o0);
              slots[k] = e < hi ? bin[at] : 0xFFFFFFFFu;
          }
#pragma unroll
          for (int k = 0; k < kEntriesPerLane; ++k) {
              recs[k].x0 = kNoRaster;
              if (slots[k] != 0xFFFFFFFFu) recs[k] = a.rast[slots[k]];
          }
#pragma unroll
          for (int k = 0; k < kEntriesPerLane; ++k) {
            const bool valid = slots[k] != 0xFFFFFFFFu;
            bool wide_box = false;
            const RastRec r = recs[k];
            // what ranks the triangle among the fragments of a pixel: its slot, or for a fan
            // triangle of a clipped primitive (rare) the slot of that primitive
            uint32_t order = slots[k];
            if (valid && order >= a.n_slots) order = a.fan_order[order - a.n_slots];
            Tri t = {0, 0, 0, 0, 0, 0};
            int32_t bx0 = 0, bx1 = 0, by0 = 0, by1 = 0;
            bool nonempty = false, is_narrow = false;
            int32_t box_px = 0;
            if (valid) {
                t = unpack_tri(r, a.wide, slots[k]);
                nonempty = pixel_box(t, cx0, cx1, cy0, cy1, bx0, bx1, by0, by1);
                is_narrow = r.d1 != kWideTri;
                box_px = mul_narrow(bx1 - bx0 + 1, by1 - by0 + 1);
            }
            // Boxes of 25 to 64 pixels: when a wave holds many of them its lanes are evenly loaded
            // and walking them lane by lane is the cheapest way; a few among small boxes would keep
            // the whole wave waiting, so they go to the queue that lane groups share.
            const bool middling = nonempty && is_narrow && box_px > kLaneBox && box_px <= kLaneBoxCrowd;
            const int32_t lane_box = __popcll(__ballot(middling)) >= kCrowd ? kLaneBoxCrowd : kLaneBox;
            if (valid) {
                const bool alone = nonempty && box_px <= lane_box && is_narrow;
                wide_box = nonempty && !alone && !(a.debug & kDebugSkipLarge);
                if (alone && !(a.debug & kDebugSkipSmall)) raster_lane(vis, t, r, order, bx0, bx1, by0, by1, tpx, tpy);
                if (a.debug & kDebugSkipAtomic) { // development census of the two raster paths
                    FrameCounters *ctr = const_cast<FrameCounters *>(a.counters);
                    const uint32_t box = (uint32_t)box_px;
                    if (wide_box) { atomicAdd(&ctr->pad[0], 1u); atomicAdd(&ctr->pad[1], box); }
                    if (alone) atomicAdd(&ctr->pad[2], box);
                }
                if (wide_box) {
                    // boxes too large for one lane are shared by all four waves after this pass
                    const uint32_t qi = atomicAdd(&s_nwq, 1u);
                    if (qi < kWideQueue) {
                        // the whole triangle, so that the second pass needs no memory access
                        WideEntry q;
                        q.x0 = t.x0; q.y0 = t.y0; q.x1 = t.x1; q.y1 = t.y1; q.x2 = t.x2; q.y2 = t.y2;
                        q.z0 = r.z0; q.dz1 = r.dz1; q.dz2 = r.dz2; q.rA = r.rA; q.order = order;
                        q.box = pack_box(bx0, bx1, by0, by1, tpx, tpy, is_narrow);
                        s_wq[qi] = q;
                        wide_box = false;
                    }
                }
            }
            // queue full: this wave walks the box on its own, 64 pixels per step
            unsigned long long lm = __ballot(wide_box);
            if (lm) {
                const WideSetup ws = wide_setup(t, is_narrow, r.rA, bx0, bx1, by0, by1);
                lm = __ballot(wide_box && !ws.misses);
                while (lm) {
                    const int src = __builtin_amdgcn_readfirstlane(__ffsll((long long)lm) - 1);
                    lm &= lm - 1;
                    raster_wide(vis, src, ws, t, r, order, (uint32_t)lane, 64u, tpx, tpy);
                }
            }
          }
        }
    }
    SXE_STAMP(2);
    __syncthreads();
    SXE_STAMP(3);
    {
        const uint32_t nq = min(s_nwq, (uint32_t)kWideQueue);
        // Middle-sized boxes: a group of 16 lanes per triangle, 16 pixels per step.  Every lane of
        // the group reads the same queue entry and prepares the edge functions for itself.
        const uint32_t gl = (uint32_t)lane & 15u, grp = (uint32_t)wave * 4u + ((uint32_t)lane >> 4);
        for (uint32_t e = grp; e < nq; e += 16u) {
            const WideEntry q = s_wq[e];
            if (!box_for_group(q.box)) continue;
            const int32_t bx0 = tpx + (int32_t)(q.box & 31u), by0 = tpy + (int32_t)((q.box >> 5) & 31u);
            const uint32_t bw = ((q.box >> 10) & 31u) + 1u, cnt = bw * (((q.box >> 15) & 31u) + 1u);
            const Tri t = {q.x0, q.y0, q.x1, q.y1, q.x2, q.y2};
            const EdgeStart es = edge_start(t, bx0, by0);
            const float rA = q.rA;
            const int32_t ax0 = -256 * es.ey0, ax1 = -256 * es.ey1, ax2 = -256 * es.ey2; // one pixel to the right
            const int32_t ay0 = 256 * es.ex0, ay1 = 256 * es.ex1, ay2 = 256 * es.ex2;    // one pixel down
            const uint32_t magic = s_magic[bw];
            for (uint32_t idx = gl; idx < cnt; idx += 16u) {
                const uint32_t row = __umul24(idx, magic) >> 16;
                const int32_t di = (int32_t)(idx - __umul24(row, bw)), dj = (int32_t)row;
                // steps are below 2^22 and pixel offsets below 2^6: 24-bit multiplies are exact
                const int32_t E0 = es.R0 + mul_narrow(ax0, di) + mul_narrow(ay0, dj);
                const int32_t E1 = es.R1 + mul_narrow(ax1, di) + mul_narrow(ay1, dj);
                const int32_t E2 = es.R2 + mul_narrow(ax2, di) + mul_narrow(ay2, dj);
                if ((E0 | E1 | E2) >= 0) {
                    const float l1 = (float)(E1 + es.b1) * rA, l2 = (float)(E2 + es.b2) * rA;
                    const float z = clamp_depth((q.z0 + l1 * q.dz1) + l2 * q.dz2);
                    depth_min(vis, (by0 + dj - tpy) * kTile + (bx0 + di - tpx), z, q.order);
                }
            }
        }
        // The rest: every wave loads the queued triangles and takes every fourth 64-pixel step of each
        for (uint32_t base = 0; base < nq; base += 64u) {
            const uint32_t e = base + (uint32_t)lane;
            const bool valid = e < nq && !box_for_group(s_wq[min(e, (uint32_t)kWideQueue - 1u)].box);
            RastRec r;
            r.x0 = r.y0 = 0; r.d1 = r.d2 = 0; r.z0 = r.dz1 = r.dz2 = r.rA = 0.0f;
            uint32_t order = 0;
            bool is_narrow = true;
            Tri t = {0, 0, 0, 0, 0, 0};
            int32_t bx0 = 0, bx1 = 0, by0 = 0, by1 = 0;
            if (valid) {
                const WideEntry q = s_wq[e];
                t.x0 = q.x0; t.y0 = q.y0; t.x1 = q.x1; t.y1 = q.y1; t.x2 = q.x2; t.y2 = q.y2;
                r.z0 = q.z0; r.dz1 = q.dz1; r.dz2 = q.dz2; r.rA = q.rA;
                order = q.order;
                is_narrow = (q.box & kBoxNarrow) != 0u;
                pixel_box(t, cx0, cx1, cy0, cy1, bx0, bx1, by0, by1);
            }
            const WideSetup ws = wide_setup(t, is_narrow, r.rA, bx0, bx1, by0, by1);
            unsigned long long lm = __ballot(valid && !ws.misses);
            while (lm) {
                const int src = __builtin_amdgcn_readfirstlane(__ffsll((long long)lm) - 1);
                lm &= lm - 1;
                raster_wide(vis, src, ws, t, r, order, (uint32_t)(wave * 64 + lane), (uint32_t)kRasterThreads, tpx, tpy);
            }
        }
    }
    __syncthreads();
    SXE_STAMP(4);
    if (a.debug & kDebugSkipResolve) {
        SXE_STAMP(7);
        return;
    }

    // ---- resolve: shade the winning triangle of every pixel once, blend over the cleared target ----
    // The covered pixels are gathered into a dense list, so that the shading loop runs with full
    // waves, and the list is in the order of the low bits of the winners' slot numbers (a counting
    // sort over 256 buckets): the pixels of one triangle, and those of its neighbours in the mesh,
    // whose records are its neighbours in memory, then sit in adjacent lanes, and a wave's gather
    // of shading records touches fewer cache lines than in pixel order (worth 1 % of the kernel;
    // which bits of the slot number make the key hardly matters).  A thread looks after pixels
    // tid, tid + 256, ... of the tile here and again in the last pass.
    uint32_t *s_bucket = s_overlay.bucket;
    s_bucket[tid] = 0u;
    __syncthreads();
    uint32_t place[kTilePixels / kRasterThreads]; // bucket | rank in the bucket << 8; ~0: not covered
#pragma unroll
    for (int k = 0; k < kTilePixels / kRasterThreads; ++k) {
        const unsigned long long key = vis[tid + k * kRasterThreads];
        place[k] = 0xFFFFFFFFu;
        if (key != kVisEmpty) {
            const uint32_t b = (uint32_t)key & 255u;
            place[k] = b | (atomicAdd(&s_bucket[b], 1u) << 8);
        }
    }
    __syncthreads();
    if (wave == 0) { // exclusive scan of the 256 counts, four per lane
        uint4 c = reinterpret_cast<uint4 *>(s_bucket)[lane];
        uint32_t sum = c.x + c.y + c.z + c.w, run = sum;
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(run, d);
            if (lane >= d) run += y;
        }
        const uint32_t before = run - sum;
        reinterpret_cast<uint4 *>(s_bucket)[lane] = make_uint4(before, before + c.x, before + c.x + c.y, before + c.x + c.y + c.z);
        if (lane == 63) s_ncov = run;
    }
    __syncthreads();
    SXE_STAMP(5);
#pragma unroll
    for (int k = 0; k < kTilePixels / kRasterThreads; ++k)
        if (place[k] != 0xFFFFFFFFu) s_cov[s_bucket[place[k] & 255u] + (place[k] >> 8)] = (uint16_t)(tid + k * kRasterThreads);
    __syncthreads();
    const uint32_t n_cov = s_ncov;
    // The records of a thread's next pixel are requested while the texels of the current one are
    // on their way: their round trip overlaps the filtering and encoding of this pixel.
    PixelWork w, w_next;
    if ((uint32_t)tid < n_cov) request_pixel(a, vis, s_cov, (uint32_t)tid, w_next);
    for (uint32_t q = (uint32_t)tid; q < n_cov; q += kRasterThreads) {
        w = w_next;
        if (q + kRasterThreads < n_cov) request_pixel(a, vis, s_cov, q + kRasterThreads, w_next);
        const PixelMid m = shade_begin<MANYTEX>(a, w, tpx, tpy, s_tex);
        shade_end<COVERAGE>(a, m, s_dec, s_enc, vis);
    }
    __syncthreads();
    // every pixel of the tile that exists in the frame and the scissor: its colour and depth, or the clear values
    for (int k = 0; k < kTilePixels / kRasterThreads; ++k) {
        const int lx = tid & (kTile - 1), ly = (tid / kTile) + k * (kRasterThreads / kTile);
        const int32_t i = tpx + lx, j = tpy + ly;
        if (i > cx1 || j < cy0 || j > cy1) continue;
        const unsigned long long v = vis[tid + k * kRasterThreads];
        const bool covered = (uint32_t)(v >> 32) != 0x3F800000u; // (a fragment's depth is below 1)
        const size_t p = (size_t)(__umul24((uint32_t)j, a.g.width) + (uint32_t)i); // (frames are at most 8192 wide and high)
        store_pixel(a, p, target_order(covered ? (uint32_t)v : a.clear_code, a.bgra), __uint_as_float((uint32_t)(v >> 32)));
        if (COVERAGE && !covered) a.out_prim[p] = 0xFFFFFFFFu;
    }
    SXE_STAMP(7);
}

// One thread per cell of a texture's table: column c in [-1, width], pair of rows j, j + 1 with j in
// [-1, height - 1], every coordinate clamped to the texture (clamp to edge, src/renderer/drawer.rs:231).
__global__ __launch_bounds__(256) void k_texture_footprints(const uint32_t *__restrict__ texels, uint32_t width, uint32_t height,
                                                            uint2 *__restrict__ cells)
{
    const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t per_row = (uint64_t)width + 2u;
    if (idx >= per_row * ((uint64_t)height + 1u)) return;
    const int c = (int)(idx % per_row) - 1, j = (int)(idx / per_row) - 1;
    const uint32_t x = (uint32_t)min(max(c, 0), (int)width - 1);
    const uint32_t y0 = (uint32_t)max(j, 0), y1 = (uint32_t)min(j + 1, (int)height - 1);
    cells[idx] = make_uint2(texels[(size_t)y0 * width + x], texels[(size_t)y1 * width + x]);
}

hipError_t launch_texture_footprints(const uint8_t *texels, uint32_t width, uint32_t height, uint2 *cells, hipStream_t s)
{
    const uint64_t n = ((uint64_t)width + 2u) * ((uint64_t)height + 1u);
    hipLaunchKernelGGL(k_texture_footprints, dim3((uint32_t)((n + 255u) / 256u)), dim3(256), 0, s,
                       reinterpret_cast<const uint32_t *>(texels), width, height, cells);
    return hipGetLastError();
}

hipError_t launch_raster_shade(const RasterArgs &a, hipStream_t s)
{
    if (a.g.n_tiles == 0) return hipSuccess;
    const bool dev = a.debug != 0u || a.tile_times != nullptr, many = a.n_textures > (uint32_t)kTexCache;
    const dim3 grid(total_tiles(a.g)), block(kRasterThreads); // one workgroup per tile of every view
#define SXE_LAUNCH(C, D, M) hipLaunchKernelGGL((k_raster_shade<C, D, M>), grid, block, 0, s, a)
    if (a.out_prim) {
        if (dev) { if (many) SXE_LAUNCH(true, true, true); else SXE_LAUNCH(true, true, false); }
        else     { if (many) SXE_LAUNCH(true, false, true); else SXE_LAUNCH(true, false, false); }
    } else {
        if (dev) { if (many) SXE_LAUNCH(false, true, true); else SXE_LAUNCH(false, true, false); }
        else     { if (many) SXE_LAUNCH(false, false, true); else SXE_LAUNCH(false, false, false); }
    }
#undef SXE_LAUNCH
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Ordered path: tiles that hold a triangle with a translucent texture
// ------------------------------------------------------------------------------------------

// texture(sampler2D, uv) with all four channels: colour decoded from sRGB before filtering, alpha
// as UNORM8 (standard.frag:12 on an Rgba8Srgb image, src/renderer/types.rs:186).
__device__ inline void filter_quad_rgba(const float *dec, const float *un, uint32_t t00, uint32_t t10, uint32_t t01, uint32_t t11,
                                        float ax, float ay, float out[4])
{
    for (int k = 0; k < 4; ++k) {
        const int sh = 8 * k;
        const float *lut = k < 3 ? dec : un;
        const float c00 = lut[(t00 >> sh) & 255], c10 = lut[(t10 >> sh) & 255];
        const float c01 = lut[(t01 >> sh) & 255], c11 = lut[(t11 >> sh) & 255];
        const float top = c00 + ax * (c10 - c00);
        const float bot = c01 + ax * (c11 - c01);
        out[k] = top + ay * (bot - top);
    }
}

// One workgroup of sixteen waves per flagged tile; a wave owns a block of 8x8 pixels, a thread one
// pixel.  The tile's entries (kBlendMaxEntries at a time: a fuller tile is taken in consecutive
// stretches of the submission order) are sorted by submission order (a record's rank, then its record
// number: the fan triangles of a clipped primitive share a rank and follow each other as the fan
// lists them) and their pixel boxes inside the tile worked out.  A wave then visits, in order,
// only the entries whose box reaches its block: coverage, depth test Less with depth write,
// shading, and BlendState::ALPHA over what the sRGB8 target holds
// (src/renderer/drawer.rs:660-663,698-705), exactly as the fixed-function pipeline does it one
// fragment at a time.  What bounds a crowded tile is the number of instructions its sixteen waves
// issue on the one compute unit: visits times instructions per visit.
constexpr int kBlendThreads = 1024;

// flags of an entry's coverage numbers: the biases that the second and third edge function contain; a wide triangle
constexpr uint32_t kEdgeBias1 = 1u, kEdgeBias2 = 2u, kEdgeWide = 4u;

template <bool COVERAGE>
__global__ __launch_bounds__(kBlendThreads) void k_raster_blend(RasterArgs a)
{
    __shared__ unsigned long long s_key[kBlendMaxEntries];
    __shared__ __attribute__((aligned(16))) uint32_t s_enc[kEncPadded];
    __shared__ float s_dec[256], s_un[256];
    constexpr uint32_t kNoBox = 0xFFFFFFFFu;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (a.counters->overflow) return;
    const uint4 launch = reinterpret_cast<const uint4 *>(a.tile_launch)[2 * blockIdx.x];
    const uint4 launch2 = reinterpret_cast<const uint4 *>(a.tile_launch)[2 * blockIdx.x + 1];
    if (!((uint32_t)__builtin_amdgcn_readfirstlane((int)launch2.w) & kTileBlend)) return;
    const uint32_t tile = (uint32_t)__builtin_amdgcn_readfirstlane((int)launch.x);
    const uint32_t view = (uint32_t)__builtin_amdgcn_readfirstlane((int)launch2.z);
#define SXE_BSTAMP(i) do { if (a.tile_times && tid == 0) a.tile_times[8 * tile + (i)] = wall_clock64(); } while (0)
    SXE_BSTAMP(0);
    a.out_color += 4 * a.image_stride * view;
    if (a.out_depth) a.out_depth += a.image_stride * view; // (null: the depth images are not kept)
    if (COVERAGE) a.out_prim += a.image_stride * view;
    for (int i = tid; i < kEncPadded; i += kBlendThreads) s_enc[i] = reinterpret_cast<const uint32_t *>(g_srgb_encode)[i];
    if (tid < 256) {
        s_dec[tid] = c_srgb_decode[tid];
        s_un[tid] = c_unorm8[tid];
    }
    const int32_t tpx = (int32_t)(tile % a.g.tiles_x) * kTile;
    const int32_t tpy = (int32_t)(tile / a.g.tiles_x + a.g.tile_row0) * kTile;
    const int32_t cx0 = tpx, cx1 = min(tpx + kTile - 1, (int32_t)a.g.width - 1);
    const int32_t cy0 = max(tpy, (int32_t)a.g.sy0), cy1 = min(tpy + kTile - 1, (int32_t)a.g.sy1 - 1);

    // The tile's entries lie in its five bin regions in the order they arrived; entry e of the
    // regions taken one after the other:
    const uint32_t cnt[kBinRegions] = {(uint32_t)__builtin_amdgcn_readfirstlane((int)launch2.y), (uint32_t)__builtin_amdgcn_readfirstlane((int)launch2.x),
                                       (uint32_t)__builtin_amdgcn_readfirstlane((int)launch.w), (uint32_t)__builtin_amdgcn_readfirstlane((int)launch.z),
                                       (uint32_t)__builtin_amdgcn_readfirstlane((int)launch.y)}; // classes 0-3, late
    const uint32_t *bin = a.bins + (size_t)(view * a.g.n_tiles + tile) * a.bl.stride;
    const uint32_t c1 = cnt[0], c2 = c1 + cnt[1], c3 = c2 + cnt[2], c4 = c3 + cnt[3], n_total = c4 + cnt[4];
    const uint32_t r0 = a.bl.off[0], r1 = a.bl.off[1] - c1, r2 = a.bl.off[2] - c2, r3 = a.bl.off[3] - c3, r4 = a.bl.off[4] - c4;
    const uint32_t n_slots = a.n_slots;
    const uint32_t *fan_order = a.fan_order;
#define SXE_ENTRY_AT(e, rec, order)                                                                      \
    do {                                                                                                 \
        rec = bin[(e) + ((e) < c1 ? r0 : (e) < c2 ? r1 : (e) < c3 ? r2 : (e) < c4 ? r3 : r4)];           \
        order = rec < n_slots ? rec : fan_order[rec - n_slots]; /* a fan triangle ranks as its primitive does */ \
    } while (0)
    uint32_t n = 0; // entries of the piece being blended
    // The tile is 4x4 blocks of 8x8 pixels and every wave owns one: a small triangle reaches one or two waves.
    // Which wave owns which block turns by one column from one row of blocks to the next: waves 0, 4, 8, 12 share an
    // execution unit, and a crowded corner of the tile should not land on one unit alone.
    const int32_t wy = wave >> 2, wx = (wave + wy) & 3;
    const int32_t tx = 8 * wx + (lane & 7), ty = 8 * wy + (lane >> 3); // in the tile
    const int32_t px = tpx + tx, py = tpy + ty;
    const int32_t sx = 256 * px + 128, sy = 256 * py + 128;
    float depth = 1.0f;
    uint32_t code = a.clear_code, prim = 0xFFFFFFFFu;
    // A fragment that has passed the depth test waits in its lane, its texels on their way, until
    // another one arrives for the same lane or the entries end: filtering and blending cost more
    // than coverage and depth do, small triangles each reach a few lanes only, and waiting lets
    // one pass of that code serve the fragments of many entries.  Only the order of a pixel's own
    // fragments matters, and a lane never holds two.
    bool pend = false;
    uint32_t pend_rec = 0u;
    float pend_l1 = 0.0f, pend_l2 = 0.0f;
    auto blend_pending = [&]() {
        if (!pend) return;
        pend = false;
        const ShadeRec h = a.shade[pend_rec];
        prim = h.prim;
        const float l1 = pend_l1, l2 = pend_l2;
        const float iw = (h.iw0 + l1 * h.diw1) + l2 * h.diw2;
        const float riw = 1.0f / iw;
        float src[4];
        if (a.shading == SXE_SHADE_VERTEX_COLOR) {
            const ColorRec c = a.color[pend_rec];
            src[0] = ((c.rw0 + l1 * c.drw1) + l2 * c.drw2) * riw;
            src[1] = ((c.gw0 + l1 * c.dgw1) + l2 * c.dgw2) * riw;
            src[2] = ((c.bw0 + l1 * c.dbw1) + l2 * c.dbw2) * riw;
            src[3] = 1.0f;
        } else {
            const float u = ((h.uw0 + l1 * h.duw1) + l2 * h.duw2) * riw;
            const float v = ((h.vw0 + l1 * h.dvw1) + l2 * h.dvw2) * riw;
            const TexelQuad tq = texel_quad(a.textures[h.tex_fan & 0xFFFFFFu], u, v);
            const uint4 fp = load_footprint(a.footprints, tq.entry);
            filter_quad_rgba(s_dec, s_un, fp.x, fp.z, fp.y, fp.w, tq.ax, tq.ay, src);
            if (a.shading == SXE_SHADE_COLOR_TIMES_TEXTURE) {
                const ColorRec c = a.color[pend_rec];
                src[0] = (((c.rw0 + l1 * c.drw1) + l2 * c.drw2) * riw) * src[0];
                src[1] = (((c.gw0 + l1 * c.dgw1) + l2 * c.dgw2) * riw) * src[1];
                src[2] = (((c.bw0 + l1 * c.dbw1) + l2 * c.dbw2) * riw) * src[2];
            } else if (a.shading == SXE_SHADE_LIT) {
                const float4 lit = reinterpret_cast<const float4 *>(&a.color[pend_rec])[2]; // (dbw2, lit_a, lit_s, -)
                src[0] = src[0] * lit.y + lit.z;
                src[1] = src[1] * lit.y + lit.z;
                src[2] = src[2] * lit.y + lit.z;
            }
        }
        // BlendState::ALPHA on an sRGB8 target: the destination is what the target holds
        const float sa = src[3], ia = 1.0f - sa;
        const float dr = s_dec[code & 255u], dg = s_dec[(code >> 8) & 255u], db = s_dec[(code >> 16) & 255u];
        const float da = s_un[code >> 24];
        const uint32_t cr = srgb_encode_lds(s_enc, src[0] * sa + dr * ia);
        const uint32_t cg = srgb_encode_lds(s_enc, src[1] * sa + dg * ia);
        const uint32_t cb = srgb_encode_lds(s_enc, src[2] * sa + db * ia);
        code = cr | (cg << 8) | (cb << 16) | (unorm8_encode(sa + da * ia) << 24);
    };
    // Sixty-four entries of the tile at a time, one per lane.  A lane whose entry reaches the
    // wave's blocks fetches the triangle and works out, once, what the coverage test of
    // any pixel needs; the wave then takes those entries in order, each one's numbers read from
    // its lane.  The fetch for the next sixty-four is under way while these are visited.
    auto fetch = [&](uint32_t base, unsigned long long &key, bool &reaches, RastRec &r) {
        key = base + (uint32_t)lane < n ? s_key[base + (uint32_t)lane] : ~0ull;
        const uint32_t box = (uint32_t)(key >> 32);
        const int32_t X0 = (int32_t)((box >> 3) & 31u), X1 = (int32_t)((box >> 19) & 31u); // block columns and rows of the box
        const int32_t Y0 = (int32_t)((box >> 11) & 31u), Y1 = (int32_t)(box >> 27);
        reaches = box != kNoBox && X0 <= wx && X1 >= wx && Y0 <= wy && Y1 >= wy;
        r = RastRec{};
        if (reaches) r = a.rast[(uint32_t)key];
    };
    const auto from_lane = [](float v, int from) { return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), from)); };
    // A tile with more entries than the sort holds is taken in pieces: the entries ranked
    // first_order .. last_order, at most kBlendMaxEntries of them, are gathered, sorted and blended,
    // then the next ranks, over what the pixels hold by then (depth, colour and the waiting
    // fragments stay in their lanes).  Submission order is all that matters, and the pieces are
    // consecutive stretches of it.
    __shared__ uint32_t s_hist[64], s_fill;
    uint32_t first_order = 0u;
    for (;;) {
        uint32_t last_order = 0xFFFFFFFFu;
        if (n_total > kBlendMaxEntries) {
            // The largest last_order such that at most kBlendMaxEntries entries rank in
            // [first_order, last_order]: the ranks from `lo` on, `width` of them, are cut into up to
            // 64 equal buckets and counted; the buckets that fit are taken whole and the search goes
            // on inside the first one that does not, six bits of the rank per pass.
            uint32_t lo = first_order, budget = kBlendMaxEntries;
            unsigned long long width = 0x100000000ull - (unsigned long long)lo;
            for (;;) {
                uint32_t shift = 0;
                while ((width >> shift) > 64ull) ++shift;
                if (tid < 64) s_hist[tid] = 0u;
                __syncthreads();
                for (uint32_t e = (uint32_t)tid; e < n_total; e += kBlendThreads) {
                    uint32_t rec, order;
                    SXE_ENTRY_AT(e, rec, order);
                    if (order >= lo && (unsigned long long)(order - lo) < width) atomicAdd(&s_hist[(order - lo) >> shift], 1u);
                }
                __syncthreads();
                uint32_t fit = 0, b = 0;
                for (; b < 64u; ++b) {
                    const uint32_t k = s_hist[b];
                    if (fit + k > budget) break;
                    fit += k;
                }
                __syncthreads(); // (the counts are cleared again in the next pass)
                if (b == 64u) { // everything from `lo` on fits
                    last_order = (uint32_t)((unsigned long long)lo + width - 1ull);
                    break;
                }
                budget -= fit;
                lo += b << shift;
                if (shift == 0u) {
                    // rank `lo` alone does not fit any more: the piece ends before it.  (Entries of one rank
                    // are the fan triangles of one clipped primitive, seven at most; should a rank ever
                    // hold more than a whole piece, it is taken as far as the sort reaches.)
                    last_order = lo > first_order ? lo - 1u : lo;
                    break;
                }
                width = min(width - ((unsigned long long)b << shift), 1ull << shift);
            }
        }
        // the piece's entries with their sort keys (rank, then record number)
        if (tid == 0) s_fill = 0u;
        __syncthreads();
        for (uint32_t e0 = 0; e0 < n_total; e0 += kBlendThreads) {
            const uint32_t e = e0 + (uint32_t)tid;
            uint32_t rec = 0, order = 0;
            bool take = false;
            if (e < n_total) {
                SXE_ENTRY_AT(e, rec, order);
                take = order >= first_order && order <= last_order;
            }
            const unsigned long long takers = __ballot(take);
            if (takers) { // one place in the list per wave and round
                uint32_t at = 0;
                const int leader = __builtin_amdgcn_readfirstlane(__ffsll((long long)takers) - 1);
                if (lane == leader) at = atomicAdd(&s_fill, (uint32_t)__popcll(takers));
                at = (uint32_t)__builtin_amdgcn_readlane((int)at, leader) + (uint32_t)__popcll(takers & ((1ull << lane) - 1ull));
                if (take && at < kBlendMaxEntries) s_key[at] = ((unsigned long long)order << 32) | rec;
            }
        }
        __syncthreads();
        n = min(s_fill, kBlendMaxEntries);
        uint32_t n_pow2 = 1;
        while (n_pow2 < n) n_pow2 <<= 1;
        for (uint32_t i = n + (uint32_t)tid; i < n_pow2; i += kBlendThreads) s_key[i] = ~0ull;
        __syncthreads();
        SXE_BSTAMP(4);
        // Bitonic network.  Thread t takes elements t, t + 1024, ...: a step whose stride is below 64
        // stays inside runs of 64 elements that one wave alone reads and writes, in program order, so
        // only the steps with longer strides need the workgroup's barrier, before and after.
        for (uint32_t k = 2; k <= n_pow2; k <<= 1)
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                if (j >= 64u) __syncthreads();
                for (uint32_t i = (uint32_t)tid; i < n_pow2; i += kBlendThreads) {
                    const uint32_t partner = i ^ j;
                    if (partner > i) {
                        const unsigned long long x = s_key[i], y = s_key[partner];
                        const bool up = (i & k) == 0;
                        if ((x > y) == up) {
                            s_key[i] = y;
                            s_key[partner] = x;
                        }
                    }
                }
                if (j >= 64u) {
                    __syncthreads();
                } else {
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
            }
        __syncthreads();

        SXE_BSTAMP(1);
        // Every entry's pixel box inside the tile takes the place of its rank in the key: the order is
        // settled, and the waves choose their entries by the box.
        for (uint32_t i = (uint32_t)tid; i < n; i += kBlendThreads) {
            const uint32_t rec = (uint32_t)s_key[i];
            const RastRec r = a.rast[rec];
            uint32_t box = kNoBox;
            if (r.x0 != kNoRaster) {
                const Tri t = unpack_tri(r, a.wide, rec);
                int32_t bx0 = 0, bx1 = 0, by0 = 0, by1 = 0;
                if (pixel_box(t, cx0, cx1, cy0, cy1, bx0, bx1, by0, by1))
                    box = (uint32_t)(bx0 - tpx) | ((uint32_t)(by0 - tpy) << 8) | ((uint32_t)(bx1 - tpx) << 16) | ((uint32_t)(by1 - tpy) << 24);
            }
            s_key[i] = ((unsigned long long)box << 32) | rec;
        }
        __syncthreads();

        // From here on the waves go their own ways, without a barrier, until the piece is done.
        unsigned long long key_next = 0ull;
        bool reaches_next = false;
        RastRec r_next;
        fetch(0u, key_next, reaches_next, r_next);
        for (uint32_t base = 0; base < n; base += 64) {
            const unsigned long long key = key_next;
            const bool reaches = reaches_next;
            const RastRec r = r_next;
            if (base + 64 < n) fetch(base + 64, key_next, reaches_next, r_next);
            unsigned long long todo = __ballot(reaches);
            if (!todo) continue;
            // A narrow triangle's biased edge functions (inside: all three >= 0) at pixel (x, y) of the
            // tile are c + a x + b y; every term is exact in 32 bits, the values at the pixels of the
            // triangle's box are the ones edge_start() gives, and a and b fit 24 bits.  A wide
            // triangle keeps its first corner and offsets in (a0, b0, c0, a1) instead.
            const uint32_t box_l = (uint32_t)(key >> 32);
            int32_t a0 = r.x0, b0 = r.y0, c0 = (int32_t)r.d1, a1 = (int32_t)r.d2, b1 = 0, c1 = 0, a2 = 0, b2 = 0, c2 = 0;
            uint32_t flags_l = kEdgeWide;
            if (reaches && r.d1 != kWideTri) {
                const Tri t = unpack_tri(r, a.wide, (uint32_t)key);
                const int32_t ox = (int32_t)(box_l & 255u), oy = (int32_t)((box_l >> 8) & 255u);
                const EdgeStart es = edge_start(t, tpx + ox, tpy + oy);
                a0 = -256 * es.ey0; b0 = 256 * es.ex0; c0 = es.R0 - mul_narrow(a0, ox) - mul_narrow(b0, oy);
                a1 = -256 * es.ey1; b1 = 256 * es.ex1; c1 = es.R1 - mul_narrow(a1, ox) - mul_narrow(b1, oy);
                a2 = -256 * es.ey2; b2 = 256 * es.ex2; c2 = es.R2 - mul_narrow(a2, ox) - mul_narrow(b2, oy);
                flags_l = (es.b1 ? kEdgeBias1 : 0u) | (es.b2 ? kEdgeBias2 : 0u);
            }
            while (todo) {
                const int from = __builtin_ctzll(todo);
                todo &= todo - 1;
                const uint32_t box = (uint32_t)__builtin_amdgcn_readlane((int)box_l, from);
                const uint32_t flags = (uint32_t)__builtin_amdgcn_readlane((int)flags_l, from);
                const int32_t A0 = __builtin_amdgcn_readlane(a0, from), B0 = __builtin_amdgcn_readlane(b0, from), C0 = __builtin_amdgcn_readlane(c0, from);
                const int32_t A1 = __builtin_amdgcn_readlane(a1, from);
                const float z0 = from_lane(r.z0, from), dz1 = from_lane(r.dz1, from), dz2 = from_lane(r.dz2, from), rA = from_lane(r.rA, from);
                const bool in_box = (tx >= (int32_t)(box & 255u)) & (tx <= (int32_t)((box >> 16) & 255u)) & (ty >= (int32_t)((box >> 8) & 255u)) &
                                    (ty <= (int32_t)(box >> 24));
                bool inside;
                float l1, l2;
                if (!(flags & kEdgeWide)) {
                    const int32_t B1 = __builtin_amdgcn_readlane(b1, from), C1 = __builtin_amdgcn_readlane(c1, from);
                    const int32_t A2 = __builtin_amdgcn_readlane(a2, from), B2 = __builtin_amdgcn_readlane(b2, from), C2 = __builtin_amdgcn_readlane(c2, from);
                    const int32_t E0 = mul_narrow(A0, tx) + (mul_narrow(B0, ty) + C0);
                    const int32_t E1 = mul_narrow(A1, tx) + (mul_narrow(B1, ty) + C1);
                    const int32_t E2 = mul_narrow(A2, tx) + (mul_narrow(B2, ty) + C2);
                    inside = (E0 | E1 | E2) >= 0;
                    l1 = (float)(E1 + (int32_t)(flags & kEdgeBias1)) * rA;
                    l2 = (float)(E2 + (int32_t)((flags >> 1) & 1u)) * rA;
                } else {
                    // edges e0 = v1->v2, e1 = v2->v0, e2 = v0->v1 with the top-left rule as a bias
                    RastRec rw;
                    rw.x0 = A0; rw.y0 = B0; rw.d1 = (uint32_t)C0; rw.d2 = (uint32_t)A1;
                    const Tri t = unpack_tri(rw, a.wide, (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)key, from));
                    const long long ex0 = t.x2 - t.x1, ey0 = t.y2 - t.y1, ex1 = t.x0 - t.x2, ey1 = t.y0 - t.y2, ex2 = t.x1 - t.x0, ey2 = t.y1 - t.y0;
                    const long long E0 = ex0 * ((long long)sy - t.y1) - ey0 * ((long long)sx - t.x1);
                    const long long E1 = ex1 * ((long long)sy - t.y2) - ey1 * ((long long)sx - t.x2);
                    const long long E2 = ex2 * ((long long)sy - t.y0) - ey2 * ((long long)sx - t.x0);
                    inside = ((E0 - edge_bias((int32_t)ex0, (int32_t)ey0)) | (E1 - edge_bias((int32_t)ex1, (int32_t)ey1)) |
                              (E2 - edge_bias((int32_t)ex2, (int32_t)ey2))) >= 0;
                    l1 = (float)E1 * rA;
                    l2 = (float)E2 * rA;
                }
                const float z = clamp_depth((z0 + l1 * dz1) + l2 * dz2);
                // below the cleared 1.0, and Comparison::Less against what is there (src/renderer/drawer.rs:700)
                const bool want = in_box & inside & (z < 1.0f) & (z < depth);
                if (!__ballot(want)) continue;
                if (__ballot(want && pend)) blend_pending();
                const uint32_t rec = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)key, from);
                if (want) {
                    depth = z;
                    pend = true;
                    pend_rec = rec;
                    pend_l1 = l1;
                    pend_l2 = l2;
                }
            }
        }
        if (last_order == 0xFFFFFFFFu) break;
        first_order = last_order + 1u;
        __syncthreads(); // every wave is done with this piece's list
    }
    blend_pending();
    // out through LDS (the sort keys are done with), so that the stores run along the rows
    __syncthreads();
    SXE_BSTAMP(2);
    if (a.tile_times && tid == 0) a.tile_times[8 * tile + 5] = n;
    uint32_t *s_out = reinterpret_cast<uint32_t *>(s_key);
    s_out[ty * kTile + tx] = code;
    s_out[kTile * kTile + ty * kTile + tx] = __float_as_uint(depth);
    if (COVERAGE) s_out[2 * kTile * kTile + ty * kTile + tx] = prim;
    __syncthreads();
    const int32_t ox = tpx + (tid & 31), oy = tpy + (tid >> 5);
    if (ox <= cx1 && oy >= cy0 && oy <= cy1) {
        const size_t p = (size_t)(__umul24((uint32_t)oy, a.g.width) + (uint32_t)ox);
        store_pixel(a, p, target_order(s_out[tid], a.bgra), __uint_as_float(s_out[kTile * kTile + tid]));
        if (COVERAGE) a.out_prim[p] = s_out[2 * kTile * kTile + tid];
    }
    SXE_BSTAMP(3);
#undef SXE_BSTAMP
#undef SXE_ENTRY_AT
}

hipError_t launch_raster_blend(const RasterArgs &a, hipStream_t s)
{
    if (a.g.n_tiles == 0 || !a.blend_scene) return hipSuccess;
    const dim3 grid(total_tiles(a.g)), block(kBlendThreads);
    if (a.out_prim)
        hipLaunchKernelGGL(k_raster_blend<true>, grid, block, 0, s, a);
    else
        hipLaunchKernelGGL(k_raster_blend<false>, grid, block, 0, s, a);
    return hipGetLastError();
}

} // namespace sxe

extern "C" int sxe_srgb_encode_host(const float *x, uint8_t *codes, size_t n)
{
    static uint32_t table[sxe::kEncPadded];
    static const bool ok = sxe::build_encode_table(table);
    if (!ok || (n && (!x || !codes))) return SXE_E_INVALID;
    for (size_t i = 0; i < n; ++i) codes[i] = (uint8_t)sxe::srgb_encode_host(table, x[i]);
    return SXE_OK;
}
