// sxe_device.h -- device helpers shared by the geometry kernels and the tile kernel: the vertex
// transform and the screen-triangle record format.
#pragma once

#include "sxe_internal.h"

namespace sxe {


// gl_Position = PVM * (pos, 1) with each row summed left to right (standard.vert:28).
__device__ inline float4 transform_vertex(const float *__restrict__ pvm, float x, float y, float z)
{
    float4 r;
    r.x = ((pvm[0] * x + pvm[4] * y) + pvm[8] * z) + pvm[12];
    r.y = ((pvm[1] * x + pvm[5] * y) + pvm[9] * z) + pvm[13];
    r.z = ((pvm[2] * x + pvm[6] * y) + pvm[10] * z) + pvm[14];
    r.w = ((pvm[3] * x + pvm[7] * y) + pvm[11] * z) + pvm[15];
    return r;
}

// ------------------------------------------------------------------------------------------
// Triangle records
// ------------------------------------------------------------------------------------------

struct Tri { // one screen triangle as the raster and resolve loops use it
    int32_t x0, y0, x1, y1, x2, y2;
};

__device__ inline bool fits16(int32_t d) { return d >= -32767 && d <= 32767; }

__device__ inline Tri unpack_tri(const RastRec &r, const WideRec *__restrict__ wide, uint32_t slot)
{
    Tri t;
    t.x0 = r.x0;
    t.y0 = r.y0;
    if (r.d1 == kWideTri) {
        const WideRec w = wide[slot];
        t.x1 = w.x1; t.y1 = w.y1; t.x2 = w.x2; t.y2 = w.y2;
    } else {
        t.x1 = r.x0 + (int32_t)(int16_t)(r.d1 & 0xFFFFu);
        t.y1 = r.y0 + ((int32_t)r.d1 >> 16);
        t.x2 = r.x0 + (int32_t)(int16_t)(r.d2 & 0xFFFFu);
        t.y2 = r.y0 + ((int32_t)r.d2 >> 16);
    }
    return t;
}

// 1 / doubled area: the reciprocal of the (positive) area converted to float.  Computed once per
// triangle by the setup stage and carried in its records.
__device__ inline float recip_area(long long area2)
{
    // (an area that fits 32 bits converts to the same float either way, in one instruction)
    const float fa = area2 == (long long)(int32_t)area2 ? (float)(int32_t)area2 : (float)area2;
    return 1.0f / fa;
}

// Inclusive pixel box of a triangle inside the rectangle [cx0,cx1] x [cy0,cy1].
// Pixel (i,j) is sampled at (256 i + 128, 256 j + 128).
__device__ inline bool pixel_box(const Tri &t, int32_t cx0, int32_t cx1, int32_t cy0, int32_t cy1, int32_t &bx0,
                                 int32_t &bx1, int32_t &by0, int32_t &by1)
{
    const int32_t xmin = min(t.x0, min(t.x1, t.x2)), xmax = max(t.x0, max(t.x1, t.x2));
    const int32_t ymin = min(t.y0, min(t.y1, t.y2)), ymax = max(t.y0, max(t.y1, t.y2));
    bx0 = max((xmin + 127) >> 8, cx0);
    bx1 = min((xmax - 128) >> 8, cx1);
    by0 = max((ymin + 127) >> 8, cy0);
    by1 = min((ymax - 128) >> 8, cy1);
    return bx0 <= bx1 && by0 <= by1;
}

// True when every edge-function value at a pixel of the triangle's own box fits 32 bits.
__device__ inline bool narrow(const Tri &t)
{
    const int32_t xmin = min(t.x0, min(t.x1, t.x2)), xmax = max(t.x0, max(t.x1, t.x2));
    const int32_t ymin = min(t.y0, min(t.y1, t.y2)), ymax = max(t.y0, max(t.y1, t.y2));
    return (xmax - xmin) < 16384 && (ymax - ymin) < 16384;
}

// Top-left rule as a bias: an edge function of exactly 0 counts only on top or left edges, so
// "inside" is E - bias >= 0 with bias 0 on those edges and 1 on the others.
__device__ inline int32_t edge_bias(int32_t ex, int32_t ey) { return (ey < 0 || (ey == 0 && ex > 0)) ? 0 : 1; }

// Product of two values that fit 24 signed bits (v_mul_i32_i24 runs at full rate, the general
// 32-bit multiply at a quarter of it).  Edge vectors and sample offsets of a narrow triangle are
// below 2^15 in magnitude, so their products are exact.
__device__ inline int32_t mul_narrow(int32_t a, int32_t b) { return __mul24(a, b); }

// Biased edge functions of a narrow triangle at the centre of pixel (i, j) of its own box:
// the sample is inside when all three are >= 0.
struct EdgeStart {
    int32_t ex0, ey0, ex1, ey1, ex2, ey2;
    int32_t b1, b2;
    int32_t R0, R1, R2;
};

// The same bias for an edge vector of a narrow triangle (components below 2^15 in magnitude): the
// sign of ey * 65536 - ex decides (ey > 0, or ey == 0 and ex <= 0: not a top or left edge).
__device__ inline int32_t edge_bias_narrow(int32_t ex, int32_t ey) { return 1 + (((int32_t)((uint32_t)ey << 16) - ex) >> 31); }

// With (X, Y) the sample relative to corner 0 and d1, d2 the other corners relative to it, the
// edge functions are E1 = dy2 X - dx2 Y (edge v2 -> v0), E2 = dx1 Y - dy1 X (edge v0 -> v1) and
// E0 = area2 - E1 - E2: the same integers as ex (sy - oy) - ey (sx - ox) edge by edge, in fewer
// operations.  Every factor is below 2^15 in magnitude for a narrow triangle and a sample of its box.
__device__ inline EdgeStart edge_start(const Tri &t, int32_t i, int32_t j)
{
    EdgeStart e;
    const int32_t dx1 = t.x1 - t.x0, dy1 = t.y1 - t.y0, dx2 = t.x2 - t.x0, dy2 = t.y2 - t.y0;
    e.ex0 = dx2 - dx1; e.ey0 = dy2 - dy1;
    e.ex1 = -dx2; e.ey1 = -dy2;
    e.ex2 = dx1; e.ey2 = dy1;
    e.b1 = edge_bias_narrow(e.ex1, e.ey1);
    e.b2 = edge_bias_narrow(e.ex2, e.ey2);
    const int32_t X = 256 * i + 128 - t.x0, Y = 256 * j + 128 - t.y0;
    const int32_t F1 = mul_narrow(dy2, X) - mul_narrow(dx2, Y);
    const int32_t F2 = mul_narrow(dx1, Y) - mul_narrow(dy1, X);
    const int32_t area2 = mul_narrow(dx1, dy2) - mul_narrow(dx2, dy1);
    e.R0 = area2 - F1 - F2 - edge_bias_narrow(e.ex0, e.ey0);
    e.R1 = F1 - e.b1;
    e.R2 = F2 - e.b2;
    return e;
}

// Does a narrow triangle cover at least one sample of its pixel box?  For boxes of at most four
// pixels.  The pixels are visited in one sequence, the step to the next one chosen between "one to
// the right" and "first of the next row" by selects: lanes with boxes of different shapes (1x4,
// 4x1, 2x2) then take the same four steps instead of the 4 x 4 a pair of nested loops would.
__device__ inline bool covers_any(const Tri &t, int32_t bx0, int32_t bx1, int32_t by0, int32_t by1)
{
    const EdgeStart e = edge_start(t, bx0, by0);
    const int32_t last = bx1 - bx0;
    const int32_t cnt = (last + 1) * (by1 - by0 + 1);
    const int32_t ax0 = -256 * e.ey0, ax1 = -256 * e.ey1, ax2 = -256 * e.ey2; // one pixel to the right
    // from the last pixel of a row to the first of the next: one down, `last` to the left
    const int32_t aw0 = 256 * e.ex0 - mul_narrow(ax0, last), aw1 = 256 * e.ex1 - mul_narrow(ax1, last),
                  aw2 = 256 * e.ex2 - mul_narrow(ax2, last);
    int32_t E0 = e.R0, E1 = e.R1, E2 = e.R2, col = 0;
    bool any = false;
#pragma unroll
    for (int32_t k = 0; k < 4; ++k) {
        any = any || (k < cnt && (E0 | E1 | E2) >= 0);
        const bool wrap = col == last;
        E0 += wrap ? aw0 : ax0;
        E1 += wrap ? aw1 : ax1;
        E2 += wrap ? aw2 : ax2;
        col = wrap ? 0 : col + 1;
    }
    return any;
}

} // namespace sxe
