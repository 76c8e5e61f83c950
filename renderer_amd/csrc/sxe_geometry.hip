// sxe_geometry.hip -- gfx950 kernels of the geometry half of the raster/shade path.
//
// Frame pipeline (all on one stream, no host round trip):
//   k_object        the rotation system's update when one is waiting, PVM = (P*V)*M per object
//                   (standard.vert:28, matrix part), a conservative whole-object visibility test;
//                   clears the frame's counters
//   k_setup         one thread per submitted triangle: vertex transform, trivial reject, perspective
//                   divide, snap, orientation, bounding box, exact coverage test for tiny boxes;
//                   writes one record per surviving triangle in submission order and files it in
//                   the bins of the (up to four) screen tiles it touches, by pixel-box size class
//   k_clip          triangles that cross the near, far or a guard-band plane: Sutherland-Hodgman
//                   and one record per triangle of the resulting fan
//   k_finish        totals and capacity check, launch order of the tiles (dearest first), and the bin
//                   entries of triangles that touch more than four tiles and of clipped fans
//   k_raster_shade  (sxe_raster.hip) one workgroup per 32x32 tile
//
// What the kernels replace in the reference is fixed-function Vulkan state selected at
// src/renderer/drawer.rs:652-705 plus the vertex shader src/data/shaders/standard.vert.
//
// Numeric contract: every float expression is evaluated as written (built with
// -ffp-contract=off); coverage is exact integer arithmetic on 24.8 fixed-point vertices.
#include "sxe_device.h"

namespace sxe {

// ------------------------------------------------------------------------------------------
// Per-object kernels
// ------------------------------------------------------------------------------------------

__device__ inline void store_model(float *models, uint32_t i, const Mat4 &R)
{
    float4 *dst = reinterpret_cast<float4 *>(models + 16 * (size_t)i);
    for (int c = 0; c < 4; ++c) dst[c] = make_float4(R.m[4 * c], R.m[4 * c + 1], R.m[4 * c + 2], R.m[4 * c + 3]);
}

// Rotation::run (src/systems/rotation.rs:26-37) for object i, `n_frames` consecutive updates of it,
// followed by to_ubo (src/components/transform.rs:69-80).  Stores transform and model matrix.
__device__ inline Mat4 rotate_object(sxe_transform *t, float *models, uint32_t i, uint64_t seed, uint32_t frame,
                                     uint32_t n_frames, float delta_ms)
{
    sxe_transform tr = t[i];
    const float delta_time = delta_ms * 0.01f;
    Quat q = {tr.rotation[0], tr.rotation[1], tr.rotation[2], tr.rotation[3]};
    for (uint32_t f = 0; f < n_frames; ++f) {
        const float r = rand01(seed, frame + f, i);
        const float ud = r * delta_time * 2.0f;
        q = mul(q, quat_from_euler_deg(2.0f * ud, 10.0f * ud, 20.0f * ud));
    }
    tr.rotation[0] = q.s; tr.rotation[1] = q.x; tr.rotation[2] = q.y; tr.rotation[3] = q.z;
    t[i] = tr;
    const Mat4 M = model_matrix(tr.position, tr.scale, q);
    store_model(models, i, M);
    return M;
}

// PVM per object, plus: is the whole object certainly invisible?  The eight corners of the mesh's
// bounding box are transformed like vertices; if all of them lie outside one plane of the view
// volume by more than a bound on the rounding error of such a transform, every vertex of the mesh
// does too, so every one of its triangles would be rejected one by one anyway.  The same holds
// for objects that project wholly above or below the scissor rows.  Skipping such objects changes
// no pixel.
// It is also the first kernel of a frame and clears the frame's counters on the way.
// Grid: x over the objects, y over the views of the launch set.
__global__ void k_object(ObjectArgs a)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, view = blockIdx.y;
    const uint32_t lin = (blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;
    for (uint32_t z = lin; z < a.n_zero; z += gridDim.x * gridDim.y * blockDim.x) a.zero[z] = 0u;
    if (lin < sizeof(FrameCounters) / 4) reinterpret_cast<uint32_t *>(a.counters)[lin] = 0u;
    if (i >= a.n || !a.objects) return;
    Mat4 PV = a.pv;
    if (a.pv_views) {
        const float4 *src = reinterpret_cast<const float4 *>(a.pv_views + 16 * (size_t)view);
        for (int c = 0; c < 4; ++c) {
            const float4 v = src[c];
            PV.m[4 * c] = v.x; PV.m[4 * c + 1] = v.y; PV.m[4 * c + 2] = v.z; PV.m[4 * c + 3] = v.w;
        }
    }
    Mat4 M;
    if (a.rot_n) {
        // the rotation system's update of this frame rides along: one launch less per frame
        // (single view only: the host launches it on its own before a set of several views)
        M = rotate_object(a.transforms, a.models_rw, i, a.rot_seed, a.rot_first, a.rot_n, a.rot_delta_ms);
    } else {
        const float4 *src = reinterpret_cast<const float4 *>(a.models + 16 * (size_t)i);
        for (int c = 0; c < 4; ++c) {
            const float4 v = src[c];
            M.m[4 * c] = v.x; M.m[4 * c + 1] = v.y; M.m[4 * c + 2] = v.z; M.m[4 * c + 3] = v.w;
        }
    }
    if (a.models_snap && view == 0) store_model(a.models_snap, i, M);
    const Mat4 R = mul(PV, M);
    const size_t vi = (size_t)view * a.n + i;
    float4 *dst = reinterpret_cast<float4 *>(a.pvm + 16 * vi);
    for (int c = 0; c < 4; ++c) dst[c] = make_float4(R.m[4 * c], R.m[4 * c + 1], R.m[4 * c + 2], R.m[4 * c + 3]);

    const MeshDev mesh = a.meshes[a.object_mesh[i]];
    const float ax = fmaxf(fabsf(mesh.lo[0]), fabsf(mesh.hi[0]));
    const float ay = fmaxf(fabsf(mesh.lo[1]), fabsf(mesh.hi[1]));
    const float az = fmaxf(fabsf(mesh.lo[2]), fabsf(mesh.hi[2]));
    float S[4]; // magnitude of the terms summed into each clip coordinate
    for (int r = 0; r < 4; ++r)
        S[r] = fabsf(R.m[r]) * ax + fabsf(R.m[4 + r]) * ay + fabsf(R.m[8 + r]) * az + fabsf(R.m[12 + r]);
    const float eps = 9.5367431640625e-7f; // 2^-20: sixteen times the unit roundoff
    const float m_near = S[2] * eps, m_far = (S[3] + S[2]) * eps, m_x = (S[3] + S[0]) * eps, m_y = (S[3] + S[1]) * eps;
    int out[6] = {0, 0, 0, 0, 0, 0};
    float ys_min = __builtin_inff(), ys_max = -__builtin_inff();
    bool w_pos = true;
    for (int k = 0; k < 8; ++k) {
        const float4 c = transform_vertex(R.m, (k & 1) ? mesh.hi[0] : mesh.lo[0], (k & 2) ? mesh.hi[1] : mesh.lo[1],
                                          (k & 4) ? mesh.hi[2] : mesh.lo[2]);
        out[0] += (c.z < -m_near);
        out[1] += (c.w - c.z < -m_far);
        out[2] += (c.w + c.x < -m_x);
        out[3] += (c.w - c.x < -m_x);
        out[4] += (c.w + c.y < -m_y);
        out[5] += (c.w - c.y < -m_y);
        w_pos = w_pos && (c.w > S[3] * eps) && (c.w > 1.0e-20f);
        const float ys = (c.y / c.w + 1.0f) * a.g.half_h;
        ys_min = fminf(ys_min, ys);
        ys_max = fmaxf(ys_max, ys);
    }
    bool vis = true;
    for (int p = 0; p < 6; ++p) vis = vis && out[p] != 8;
    // scissor rows: only when every corner is in front of the eye (then the projection is bounded)
    if (w_pos && (ys_max < (float)a.g.sy0 - 2.0f || ys_min > (float)a.g.sy1 + 2.0f)) vis = false;
    if (mesh.n_tri == 0) vis = false;
    a.visible[vi] = vis ? 1 : 0;
}

// Transform::to_ubo for every object (src/components/transform.rs:69-80, loop at src/main.rs:237-244).
__global__ void k_transform_to_model(const sxe_transform *__restrict__ t, float *__restrict__ models, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const sxe_transform tr = t[i];
    const Quat q = {tr.rotation[0], tr.rotation[1], tr.rotation[2], tr.rotation[3]};
    store_model(models, i, model_matrix(tr.position, tr.scale, q));
}

// Rotation::run (src/systems/rotation.rs:26-37), `n_frames` consecutive updates of it, followed
// by to_ubo.
__global__ void k_rotation_system(sxe_transform *t, float *models, uint32_t n,
                                  uint64_t seed, uint32_t frame, uint32_t n_frames, float delta_ms)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    rotate_object(t, models, i, seed, frame, n_frames, delta_ms);
}

// ------------------------------------------------------------------------------------------
// Vertex stage
// ------------------------------------------------------------------------------------------

// Stand-alone vertex kernel over SoA position streams: each lane reads four consecutive vertices
// with three 128-bit loads and writes four clip positions.  The frame pipeline transforms
// vertices inside k_setup instead (same function); this kernel serves sxe_read_clip_positions.
__global__ __launch_bounds__(256) void k_vertex(const float *__restrict__ px, const float *__restrict__ py,
                                                const float *__restrict__ pz, const MeshDev *__restrict__ meshes,
                                                const VBlockDesc *__restrict__ vblocks,
                                                const float *__restrict__ pvm_all, float4 *__restrict__ clip)
{
    const VBlockDesc vb = vblocks[blockIdx.x];
    const MeshDev mesh = meshes[vb.mesh];
    const float *pvm = pvm_all + 16 * (size_t)vb.object;
    const uint32_t v0 = vb.vtx_start + 4 * threadIdx.x;
    if (v0 >= mesh.n_vtx) return;
    const size_t src = (size_t)mesh.vtx_base + v0; // vtx_base and v0 are multiples of 4
    const float4 X = *reinterpret_cast<const float4 *>(px + src);
    const float4 Y = *reinterpret_cast<const float4 *>(py + src);
    const float4 Z = *reinterpret_cast<const float4 *>(pz + src);
    const float xs[4] = {X.x, X.y, X.z, X.w}, ys[4] = {Y.x, Y.y, Y.z, Y.w}, zs[4] = {Z.x, Z.y, Z.z, Z.w};
    float4 *dst = clip + (size_t)vb.clip_base + v0;
    for (int k = 0; k < 4; ++k)
        if (v0 + k < mesh.n_vtx) dst[k] = transform_vertex(pvm, xs[k], ys[k], zs[k]);
}

// ------------------------------------------------------------------------------------------
// Triangle setup
// ------------------------------------------------------------------------------------------

struct ClipVert { // clip-space position + attributes: x y z w u v r g b
    float c[9];
};

__device__ inline float guard_dist(const float *c, int plane)
{
    const float x = c[0], y = c[1], z = c[2], w = c[3];
    switch (plane) {
    case 0: return z;
    case 1: return w - z;
    case 2: return kGuard * w + x;
    case 3: return kGuard * w - x;
    case 4: return kGuard * w + y;
    default: return kGuard * w - y;
    }
}

// Sutherland-Hodgman against near (0 <= z), far (z <= w) and the four guard-band planes, in that
// order.  A new corner is always interpolated from the inside end of its edge so that triangles
// sharing the edge produce identical corners.  Returns the corner count (0 if nothing is left).
__device__ int clip_polygon(ClipVert *poly, ClipVert *tmp, int n)
{
    for (int plane = 0; plane < 6; ++plane) {
        int m = 0;
        for (int i = 0; i < n; ++i) {
            const ClipVert &a = poly[i];
            const ClipVert &b = poly[(i + 1 == n) ? 0 : i + 1];
            const float da = guard_dist(a.c, plane), db = guard_dist(b.c, plane);
            if (da >= 0.0f) {
                tmp[m++] = a;
                if (!(db >= 0.0f)) {
                    const float t = da / (da - db);
                    for (int k = 0; k < 9; ++k) tmp[m].c[k] = a.c[k] + t * (b.c[k] - a.c[k]);
                    ++m;
                }
            } else if (db >= 0.0f) {
                const float t = db / (db - da);
                for (int k = 0; k < 9; ++k) tmp[m].c[k] = b.c[k] + t * (a.c[k] - b.c[k]);
                ++m;
            }
        }
        n = m;
        for (int i = 0; i < n; ++i) poly[i] = tmp[i];
        if (n < 3) return 0;
    }
    return n;
}

struct ScreenTri {
    Tri t;                      // oriented so that the doubled area is > 0
    int32_t px0, px1, py0, py1; // inclusive pixel box, clamped to the frame and the scissor rows
    float zn0, zn1, zn2;        // z/w per corner (after reorientation)
    float rw0, rw1, rw2;        // 1/w per corner
    float rA;                   // 1 / float(doubled area of t)
    bool swapped;               // corners 1 and 2 were exchanged
};

// Everything about one projected vertex that does not depend on the triangle it is used in.
struct PVert {
    int32_t x, y;   // window position, 24.8 fixed point
    float zn, rw;   // z/w and 1/w
    uint32_t mask;  // kOut* bits
};
// bits 0-5: outside a plane of the view volume (near, far, -x, +x, -y, +y); bits 6-9: outside a
// guard-band plane; bit 10: cannot be projected (w <= 0 or far outside the fixed-point range)
constexpr uint32_t kOutVolume = 0x3Fu, kOutGuard = 0x3C3u, kOutUnusable = 0x400u;

// Clip tests, perspective divide, viewport map and snap of one vertex.
__device__ inline PVert project_vertex(const float4 &c, const FrameGeom &g)
{
    PVert p;
    const float gw = kGuard * c.w;
    uint32_t m = 0;
    if (!(c.z >= 0.0f)) m |= 1u;
    if (!(c.w - c.z >= 0.0f)) m |= 2u;
    if (!(c.w + c.x >= 0.0f)) m |= 4u;
    if (!(c.w - c.x >= 0.0f)) m |= 8u;
    if (!(c.w + c.y >= 0.0f)) m |= 16u;
    if (!(c.w - c.y >= 0.0f)) m |= 32u;
    if (!(gw + c.x >= 0.0f)) m |= 64u;
    if (!(gw - c.x >= 0.0f)) m |= 128u;
    if (!(gw + c.y >= 0.0f)) m |= 256u;
    if (!(gw - c.y >= 0.0f)) m |= 512u;
    const float rw = 1.0f / c.w;
    const float xs = ((c.x * rw + 1.0f) * g.half_w) * 256.0f, ys = ((c.y * rw + 1.0f) * g.half_h) * 256.0f;
    const float lim = 1073741824.0f;
    const bool usable = (c.w > 0.0f) && (fabsf(xs) <= lim) && (fabsf(ys) <= lim);
    if (!usable) m |= kOutUnusable;
    p.x = usable ? (int32_t)rintf(xs) : 0;
    p.y = usable ? (int32_t)rintf(ys) : 0;
    p.zn = c.z * rw;
    p.rw = rw;
    p.mask = m;
    return p;
}

// Orientation and bounding box of a triangle of projected vertices.  False = nothing to draw.
__device__ inline bool setup_screen_tri(const PVert &v0, const PVert &v1, const PVert &v2, const FrameGeom &g,
                                        ScreenTri &s)
{
    if ((v0.mask | v1.mask | v2.mask) & kOutUnusable) return false;
    Tri t;
    t.x0 = v0.x; t.y0 = v0.y;
    t.x1 = v1.x; t.y1 = v1.y;
    t.x2 = v2.x; t.y2 = v2.y;
    const long long area2 = (long long)(t.x1 - t.x0) * (long long)(t.y2 - t.y0) - (long long)(t.x2 - t.x0) * (long long)(t.y1 - t.y0);
    if (area2 == 0) return false;
    const float rw0 = v0.rw;
    float zn1 = v1.zn, zn2 = v2.zn, r1 = v1.rw, r2 = v2.rw;
    s.swapped = area2 < 0;
    if (s.swapped) { // no face culling (Rasterizer::FILL): reorient instead
        int32_t q;
        q = t.x1; t.x1 = t.x2; t.x2 = q;
        q = t.y1; t.y1 = t.y2; t.y2 = q;
        float f;
        f = zn1; zn1 = zn2; zn2 = f;
        f = r1; r1 = r2; r2 = f;
    }
    if (!pixel_box(t, 0, (int32_t)g.width - 1, (int32_t)g.sy0, (int32_t)g.sy1 - 1, s.px0, s.px1, s.py0, s.py1)) return false;
    s.t = t;
    s.rA = recip_area(area2 < 0 ? -area2 : area2);
    s.zn0 = v0.zn; s.zn1 = zn1; s.zn2 = zn2;
    s.rw0 = rw0; s.rw1 = r1; s.rw2 = r2;
    return true;
}

struct Attr3 {
    float a0, a1, a2;
};

// Builds the two records every screen triangle has.  u, v are the per-corner texture coordinates
// already multiplied by the corner's 1/w, in the original corner order; corners 1 and 2 are
// exchanged when the triangle was reoriented.
__device__ inline void make_records(const SetupArgs &a, const ScreenTri &s, uint32_t prim, int32_t texture, Attr3 u, Attr3 v,
                                    RastRec &r, ShadeRec &h)
{
    r.x0 = s.t.x0; r.y0 = s.t.y0;
    const int32_t dx1 = s.t.x1 - s.t.x0, dy1 = s.t.y1 - s.t.y0, dx2 = s.t.x2 - s.t.x0, dy2 = s.t.y2 - s.t.y0;
    const bool packed = narrow(s.t); // (then every offset is below 2^14 in magnitude)
    const uint32_t d1 = ((uint32_t)dx1 & 0xFFFFu) | ((uint32_t)dy1 << 16), d2 = ((uint32_t)dx2 & 0xFFFFu) | ((uint32_t)dy2 << 16);
    r.d1 = packed ? d1 : kWideTri; // (then the corners are in the slot's WideRec: wide_record())
    r.d2 = packed ? d2 : 0u;
    r.z0 = s.zn0; r.dz1 = s.zn1 - s.zn0; r.dz2 = s.zn2 - s.zn0;
    r.rA = s.rA;
    if (s.swapped) {
        float t;
        t = u.a1; u.a1 = u.a2; u.a2 = t;
        t = v.a1; v.a1 = v.a2; v.a2 = t;
    }
    h.iw0 = s.rw0; h.diw1 = s.rw1 - s.rw0; h.diw2 = s.rw2 - s.rw0;
    if (a.shading != SXE_SHADE_VERTEX_COLOR) {
        const float uw0 = u.a0, uw1 = u.a1, uw2 = u.a2;
        const float vw0 = v.a0, vw1 = v.a1, vw2 = v.a2;
        h.uw0 = uw0; h.duw1 = uw1 - uw0; h.duw2 = uw2 - uw0;
        h.vw0 = vw0; h.dvw1 = vw1 - vw0; h.dvw2 = vw2 - vw0;
    } else {
        h.uw0 = h.duw1 = h.duw2 = h.vw0 = h.dvw1 = h.dvw2 = 0.0f;
    }
    h.prim = prim;
    h.fan_or_ra = __float_as_uint(s.rA);
    h.tex_fan = (uint32_t)texture & 0xFFFFFFu;
}

// The corners of a triangle that does not fit the packed form.
__device__ inline WideRec wide_record(const ScreenTri &s)
{
    WideRec w;
    w.x1 = s.t.x1; w.y1 = s.t.y1; w.x2 = s.t.x2; w.y2 = s.t.y2;
    return w;
}

// SXE_SHADE_LIT: the two lighting terms of a triangle, from the world positions of its corners
// (p0, p1, p2 in object space, m the object's column-major model matrix): ambient + diffuse and
// specular, the face's normal turned towards the viewer.  Every product and sum is rounded on its
// own, in the order written, so that a CPU restatement gives the same bits.
__device__ inline float2 light_terms(const float *m, const float *p0, const float *p1, const float *p2, const LightDev &L)
{
    float w[3][3];
    const float *p[3] = {p0, p1, p2};
    for (int k = 0; k < 3; ++k)
        for (int r = 0; r < 3; ++r) w[k][r] = ((m[r] * p[k][0] + m[4 + r] * p[k][1]) + m[8 + r] * p[k][2]) + m[12 + r];
    const float e1x = w[1][0] - w[0][0], e1y = w[1][1] - w[0][1], e1z = w[1][2] - w[0][2];
    const float e2x = w[2][0] - w[0][0], e2y = w[2][1] - w[0][1], e2z = w[2][2] - w[0][2];
    float nx = e1y * e2z - e1z * e2y, ny = e1z * e2x - e1x * e2z, nz = e1x * e2y - e1y * e2x;
    const float len = sqrtf((nx * nx + ny * ny) + nz * nz);
    if (!(len > 0.0f) || !(len <= 3.0e38f)) return make_float2(L.ambient, 0.0f);
    nx = nx / len; ny = ny / len; nz = nz / len;
    const float facing = (nx * L.vx + ny * L.vy) + nz * L.vz;
    if (facing < 0.0f) { nx = -nx; ny = -ny; nz = -nz; }
    float ndl = (nx * L.lx + ny * L.ly) + nz * L.lz;
    float ndh = (nx * L.hx + ny * L.hy) + nz * L.hz;
    if (!(ndl > 0.0f)) ndl = 0.0f;
    if (!(ndh > 0.0f)) ndh = 0.0f;
    float spec = ndh;
    for (uint32_t k = 0; k < L.shininess_log2; ++k) spec = spec * spec;
    return make_float2(L.ambient + L.diffuse * ndl, L.specular * spec);
}

// The colour planes (scenes shaded with vertex colours): cr/cg/cb are the plain per-corner colours
// in the original corner order.
__device__ inline ColorRec color_record(const ScreenTri &s, Attr3 cr, Attr3 cg, Attr3 cb, float2 lit)
{
    if (s.swapped) {
        float t;
        t = cr.a1; cr.a1 = cr.a2; cr.a2 = t;
        t = cg.a1; cg.a1 = cg.a2; cg.a2 = t;
        t = cb.a1; cb.a1 = cb.a2; cb.a2 = t;
    }
    const float r0 = cr.a0 * s.rw0, r1 = cr.a1 * s.rw1, r2 = cr.a2 * s.rw2;
    const float g0 = cg.a0 * s.rw0, g1 = cg.a1 * s.rw1, g2 = cg.a2 * s.rw2;
    const float b0 = cb.a0 * s.rw0, b1 = cb.a1 * s.rw1, b2 = cb.a2 * s.rw2;
    ColorRec c;
    c.rw0 = r0; c.drw1 = r1 - r0; c.drw2 = r2 - r0;
    c.gw0 = g0; c.dgw1 = g1 - g0; c.dgw2 = g2 - g0;
    c.bw0 = b0; c.dbw1 = b1 - b0; c.dbw2 = b2 - b0;
    c.lit_a = lit.x;
    c.lit_s = lit.y;
    c.pad2 = 0.0f;
    return c;
}

// All records of one screen triangle, straight to slot `rec`.
__device__ inline void write_records(const SetupArgs &a, uint32_t rec, const ScreenTri &s, uint32_t prim,
                                     int32_t texture, Attr3 u, Attr3 v, Attr3 cr, Attr3 cg, Attr3 cb, float2 lit)
{
    RastRec r;
    ShadeRec h;
    make_records(a, s, prim, texture, u, v, r, h);
    a.rast[rec] = r;
    a.shade[rec] = h;
    if (!narrow(s.t)) a.wide[rec] = wide_record(s);
    if (a.shading != SXE_SHADE_TEXTURE) a.color[rec] = color_record(s, cr, cg, cb, lit);
}

// One atomic per tile of the box: used for triangles that span several tiles and for fan triangles.
__device__ inline void count_box(const SetupArgs &a, uint32_t view, const ScreenTri &s, bool translucent)
{
    const int32_t tx0 = s.px0 / kTile, tx1 = s.px1 / kTile;
    const int32_t ty0 = s.py0 / kTile - (int32_t)a.g.tile_row0, ty1 = s.py1 / kTile - (int32_t)a.g.tile_row0;
    TileCtr *tiles = a.tiles + (size_t)view * a.g.n_tiles;
    for (int32_t ty = ty0; ty <= ty1; ++ty)
        for (int32_t tx = tx0; tx <= tx1; ++tx) {
            TileCtr &t = tiles[(uint32_t)ty * a.g.tiles_x + (uint32_t)tx];
            atomicAdd(&t.multi, 1u);
            if (translucent) atomicOr(&t.flags, kTileBlend);
        }
}

// Does a triangle with this texture need its fragments blended in order?
__device__ inline bool translucent_texture(const SetupArgs &a, int32_t texture)
{
    return a.blend_scene && a.shading != SXE_SHADE_VERTEX_COLOR && (a.textures[texture].flags & kTexTranslucent);
}

// One thread per submitted triangle; a workgroup covers 256 consecutive triangles of one object
// and owns 256 consecutive record slots, 64 per wave, which each wave fills from the front in
// submission order.  Slot numbers therefore order triangles as they were submitted, and that is
// what makes "first primitive wins a depth tie" (Comparison::Less, src/renderer/drawer.rs:700) fall
// out of a (depth | slot) minimum.  (Filling the 256 slots from the front across the waves packs
// the records a little tighter, which the raster kernel gains 2 % from, but the waves then have
// to meet once more before any of them knows its slots, which costs this kernel 4 %.)
// TEX: texture shading only, so there are no colour planes to build or stage.
// A 16-byte store marked as streaming.  The records are written once and read a kernel later,
// 240 MB of them per frame: stored the ordinary way they fill the L2 on their way out, and the
// L1s wait longer for every other request (k_setup 106.6 -> 99.6 us, and the tile kernel, which
// finds them in memory either way, 104.2 -> 100.7 us).  The same hint on the 4-byte bin entries
// gains setup what it costs the tile kernel; on the tile kernel's loads of records or textures it
// costs 2 to 20 %.
__device__ inline void stream_store(uint4 *p, uint4 v)
{
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, reinterpret_cast<u32x4 *>(p));
}
template <bool TEX>
struct SetupShared {
    static constexpr int kStageWords = TEX ? 5 : 8; // 16-byte words per record: raster 2, shading 3, colour 3
    union {
        // first the projected vertices of the work item's triangles: position, (z/w, 1/w), (u/w, v/w), flags
        struct {
            int2 xy[kVtxCache];
            float2 zr[kVtxCache];
            float2 uv[kVtxCache];
            uint32_t mask[kVtxCache];
        } vtx;
        // then the finished records of every wave, in its own order, on their way to memory
        uint4 stage[kSetupThreads * kStageWords];
    };
};

template <bool TEX>
__device__ inline void setup_block(const SetupArgs &a, const uint32_t block, const uint32_t view, SetupShared<TEX> &sh)
{
    int2 *s_xy = sh.vtx.xy;
    float2 *s_zr = sh.vtx.zr, *s_uv = sh.vtx.uv;
    uint32_t *s_mask = sh.vtx.mask;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const BlockDesc bd = a.blocks[block];
    const uint32_t t = bd.tri_start + tid;
    const bool valid = t < bd.n_tri;
    // Triangles that are neighbours in the index list share most of their vertices: when the
    // vertex range of these 256 triangles fits the cache, every vertex is projected once.
    const bool cached = bd.vtx_n <= (uint32_t)kVtxCache;
    const uint32_t vblock = view * a.n_blocks + block; // work item number over all views: names its record slots
    const size_t vobject = (size_t)view * a.n_objects + bd.object;
    uint32_t *wave_count = a.blk_count + (size_t)vblock * (kSetupThreads / 64); // survivors of each wave of this work item
    // (the flag's aligned word: a uniform word is a scalar load, a byte is a vector load)
    const uint32_t visible_word = reinterpret_cast<const uint32_t *>(a.visible)[vobject >> 2];
    if (!((visible_word >> (8u * ((uint32_t)vobject & 3u))) & 0xFFu)) {
        if (tid < kSetupThreads / 64) wave_count[tid] = 0;
        return;
    }
    const float *pvm = a.pvm + 16 * vobject;
    const bool translucent = translucent_texture(a, bd.texture); // (the same for the whole work item)
    // (asked for before the vertex phase: the reply is not on the critical path then; a lane
    // beyond the object's last triangle goes through the motions with that triangle, which costs
    // nothing, whereas defaults for everything it would otherwise leave unset cost every wave
    // twenty moves)
    const uint32_t *ip = a.indices + bd.idx_base + 3 * (size_t)min(t, bd.n_tri - 1u);
    const uint32_t i0 = ip[0], i1 = ip[1], i2 = ip[2];
    if (cached) {
        for (uint32_t i = (uint32_t)tid; i < bd.vtx_n; i += kSetupThreads) {
            const sxe_vertex &w = a.vertices[bd.vtx_base + bd.vtx_lo + i];
            const PVert p = project_vertex(transform_vertex(pvm, w.pos[0], w.pos[1], w.pos[2]), a.g);
            s_xy[i] = make_int2(p.x, p.y);
            s_zr[i] = make_float2(p.zn, p.rw);
            s_uv[i] = make_float2(w.uv[0] * p.rw, w.uv[1] * p.rw);
            s_mask[i] = p.mask;
        }
        __syncthreads();
    }
    if ((a.debug & kDebugSkipRaster) && (a.debug & kDebugSkipResolve)) { // development: stop after the vertex phase
        if (tid < kSetupThreads / 64) wave_count[tid] = 0;
        return;
    }

    // the triangle's corners; all three outside one plane of the view volume -> nothing visible;
    // one outside a guard-band, near or far plane -> the clipper's
    bool candidate = false, clip = false;
    PVert v0, v1, v2;
    Attr3 uw, vw;
    {
        if (cached) {
            const uint32_t k0 = i0 - bd.vtx_lo, k1 = i1 - bd.vtx_lo, k2 = i2 - bd.vtx_lo;
            const int2 q0 = s_xy[k0], q1 = s_xy[k1], q2 = s_xy[k2];
            const float2 z0 = s_zr[k0], z1 = s_zr[k1], z2 = s_zr[k2];
            const float2 t0 = s_uv[k0], t1 = s_uv[k1], t2 = s_uv[k2];
            v0 = {q0.x, q0.y, z0.x, z0.y, s_mask[k0]};
            v1 = {q1.x, q1.y, z1.x, z1.y, s_mask[k1]};
            v2 = {q2.x, q2.y, z2.x, z2.y, s_mask[k2]};
            uw = {t0.x, t1.x, t2.x};
            vw = {t0.y, t1.y, t2.y};
        } else {
            const sxe_vertex &w0 = a.vertices[bd.vtx_base + i0], &w1 = a.vertices[bd.vtx_base + i1], &w2 = a.vertices[bd.vtx_base + i2];
            v0 = project_vertex(transform_vertex(pvm, w0.pos[0], w0.pos[1], w0.pos[2]), a.g);
            v1 = project_vertex(transform_vertex(pvm, w1.pos[0], w1.pos[1], w1.pos[2]), a.g);
            v2 = project_vertex(transform_vertex(pvm, w2.pos[0], w2.pos[1], w2.pos[2]), a.g);
            uw = {w0.uv[0] * v0.rw, w1.uv[0] * v1.rw, w2.uv[0] * v2.rw};
            vw = {w0.uv[1] * v0.rw, w1.uv[1] * v1.rw, w2.uv[1] * v2.rw};
        }
        const bool rej = (v0.mask & v1.mask & v2.mask & kOutVolume) != 0;
        const bool inside = ((v0.mask | v1.mask | v2.mask) & kOutGuard) == 0;
        candidate = valid && !rej && inside;
        clip = valid && !rej && !inside;
    }
    // The projected vertices have been read: from here on the same LDS stages the finished
    // records.  A wave keeps to itself after this barrier: it has its own slots and its own piece
    // of the staging area.
    __syncthreads();
    constexpr int kStageWords = SetupShared<TEX>::kStageWords;
    const unsigned long long below = (1ull << lane) - 1ull;
    uint4 *sr = sh.stage + wave * (kStageWords * 64), *ss = sr + 2 * 64, *sc = sr + 5 * 64; // raster, shading, colour records
    const unsigned long long clip_mask = __ballot(clip);
    const uint32_t wslot0 = (vblock * (kSetupThreads / 64) + (uint32_t)wave) * 64u; // the wave's first slot
    const uint32_t prim = bd.prim_base + t;
    const uint32_t tile0 = view * a.g.n_tiles; // the view's first tile in the per-tile tables and the bins

    bool fast = false;
    if (candidate) {
        ScreenTri st;
        if (setup_screen_tri(v0, v1, v2, a.g, st)) {
            // a triangle whose box holds a few pixels and contains none of their samples draws nothing
            const int32_t bw = st.px1 - st.px0 + 1, bh = st.py1 - st.py0 + 1;
            const bool is_narrow = narrow(st.t);
            if (!(bw * bh <= 4 && is_narrow && !covers_any(st.t, st.px0, st.px1, st.py0, st.py1))) {
                fast = true;
                // Rank among the wave's survivors (clipped triangles keep a slot too): the lanes that
                // are here are exactly the ones that draw without the clipper.
                const unsigned long long fast_mask = __ballot(true);
                const uint32_t rank = (uint32_t)__popcll((fast_mask | clip_mask) & below);
                const uint32_t slot = wslot0 + rank;

                // Bin entries.  A triangle that lies in one tile, or touches up to four, reserves an
                // entry in the region of its size class in each of those bins, one atomic per (tile,
                // size class) group of the wave: neighbouring triangles fall into the same tiles, and
                // a reservation that many lanes make on one counter one after the other is slow to
                // come back (grouping only lanes that follow one another, which needs no loop over the
                // distinct counters, made a third more atomics and the kernel a quarter slower).  The
                // reply is needed only after the records have been built.
                const int32_t tx0 = st.px0 / kTile, ty0 = st.py0 / kTile;
                const uint32_t cols_m1 = (uint32_t)(st.px1 / kTile - tx0), rows_m1 = (uint32_t)(st.py1 / kTile - ty0);
                const uint32_t n_touched = (cols_m1 + 1u) * (rows_m1 + 1u);
                const bool few = n_touched <= 4u; // (one tile included)
                const uint32_t part = size_class((uint32_t)(bw * bh), is_narrow);
                const uint32_t first = tile0 + (uint32_t)(ty0 - (int32_t)a.g.tile_row0) * a.g.tiles_x + (uint32_t)tx0;
                uint32_t raw4[4] = {0, 0, 0, 0};   // the reservation of the lane's group, valid in its leader lane
                uint32_t who4[4] = {0, 0, 0, 0};   // leader lane | rank inside the group << 8
                {
                    uint32_t cx = 0, cy = 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const bool has = few && (uint32_t)j < n_touched;
                        const uint32_t counter = 4u * (first + cy * a.g.tiles_x + cx) + part;
                        who4[j] = (uint32_t)lane;
                        uint32_t count_j = 0;
                        unsigned long long pending = __ballot(has);
                        while (pending) {
                            const int leader = __builtin_amdgcn_readfirstlane(__ffsll((long long)pending) - 1);
                            const uint32_t lk = __builtin_amdgcn_readlane(counter, leader);
                            const bool mine = has && counter == lk;
                            const unsigned long long same = __ballot(mine);
                            if (mine) {
                                who4[j] = (uint32_t)leader | ((uint32_t)__popcll(same & below) << 8);
                                count_j = (uint32_t)__popcll(same);
                            }
                            pending &= ~same;
                        }
                        if (has && (uint32_t)lane == (who4[j] & 255u)) {
                            raw4[j] = atomicAdd(&a.tiles[counter >> 2].cnt[counter & 3u], count_j);
                            if (translucent) atomicOr(&a.tiles[counter >> 2].flags, kTileBlend);
                        }
                        if (++cx > cols_m1) { cx = 0; ++cy; }
                    }
                }
                {
                    // Triangles over more than four tiles are only counted here; their slots go on a
                    // list that k_finish spreads evenly over waves of its own (one list position per
                    // wave of this kernel would leave a wave full of such triangles with all of their
                    // entries to file).
                    const bool walker = !few;
                    if (walker) count_box(a, view, st, translucent);
                    const unsigned long long wm = __ballot(walker);
                    if (wm) {
                        const int leader = __builtin_amdgcn_readfirstlane(__ffsll((long long)wm) - 1);
                        uint32_t at = 0;
                        if (lane == leader) at = atomicAdd(&a.counters->walk_count, (uint32_t)__popcll(wm));
                        at = (uint32_t)__builtin_amdgcn_readlane((int)at, leader);
                        if (walker) a.walk_list[at + (uint32_t)__popcll(wm & below)] = slot;
                    }
                }

                // The records go to memory through LDS: a lane storing its own 32- and 48-byte
                // records would send the L2 one 16-byte request per lane and instruction; laid out
                // in slot order first, the wave stores them as whole cache lines.
                RastRec r;
                ShadeRec h;
                make_records(a, st, prim, bd.texture, uw, vw, r, h);
                sr[2 * rank] = make_uint4((uint32_t)r.x0, (uint32_t)r.y0, r.d1, r.d2);
                sr[2 * rank + 1] = make_uint4(__float_as_uint(r.z0), __float_as_uint(r.dz1), __float_as_uint(r.dz2), __float_as_uint(r.rA));
                ss[3 * rank] = make_uint4(__float_as_uint(h.iw0), __float_as_uint(h.diw1), __float_as_uint(h.diw2), __float_as_uint(h.uw0));
                ss[3 * rank + 1] = make_uint4(__float_as_uint(h.duw1), __float_as_uint(h.duw2), __float_as_uint(h.vw0), __float_as_uint(h.dvw1));
                ss[3 * rank + 2] = make_uint4(__float_as_uint(h.dvw2), h.prim, h.fan_or_ra, h.tex_fan);
                if (!TEX) {
                    // (the indices are read again rather than kept in registers all the way down here)
                    const uint32_t *ip = a.indices + bd.idx_base + 3 * (size_t)t;
                    const sxe_vertex &w0 = a.vertices[bd.vtx_base + ip[0]], &w1 = a.vertices[bd.vtx_base + ip[1]],
                                     &w2 = a.vertices[bd.vtx_base + ip[2]];
                    float2 lit = make_float2(0.0f, 0.0f);
                    if (a.shading == SXE_SHADE_LIT) lit = light_terms(a.models + 16 * (size_t)bd.object, w0.pos, w1.pos, w2.pos, a.light);
                    const ColorRec c = color_record(st, {w0.color[0], w1.color[0], w2.color[0]}, {w0.color[1], w1.color[1], w2.color[1]},
                                                    {w0.color[2], w1.color[2], w2.color[2]}, lit);
                    sc[3 * rank] = make_uint4(__float_as_uint(c.rw0), __float_as_uint(c.drw1), __float_as_uint(c.drw2), __float_as_uint(c.gw0));
                    sc[3 * rank + 1] = make_uint4(__float_as_uint(c.dgw1), __float_as_uint(c.dgw2), __float_as_uint(c.bw0), __float_as_uint(c.dbw1));
                    sc[3 * rank + 2] = make_uint4(__float_as_uint(c.dbw2), __float_as_uint(c.lit_a), __float_as_uint(c.lit_s), 0u);
                }
                if (!is_narrow) a.wide[slot] = wide_record(st);

                // File the triangle: its slot number goes to the reserved position of the size
                // class's region in the bin of every tile it touches.
                {
                    const uint32_t cap = pick_region(a.bl.cap, part), off = pick_region(a.bl.off, part);
                    uint32_t cx = 0, cy = 0;
                    bool over = false; // (then the frame is drawn again with larger regions)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const bool has = few && (uint32_t)j < n_touched;
                        const uint32_t at = (uint32_t)__shfl((int)raw4[j], (int)(who4[j] & 255u)) + (who4[j] >> 8);
                        if (has && at < cap) a.bins[(first + cy * a.g.tiles_x + cx) * a.bl.stride + off + at] = slot;
                        over |= has && at >= cap;
                        if (++cx > cols_m1) { cx = 0; ++cy; }
                    }
                    if (over) atomicOr(&a.counters->overflow, kOverflowBins);
                }
            }
        }
    }

    const unsigned long long alive = __ballot(fast || clip);
    const uint32_t n_alive = (uint32_t)__popcll(alive);
    if (lane == 0) wave_count[wave] = n_alive;
    // ---- triangles cut by the near/far/guard planes keep their slot (it ranks them in submission
    // order) and are queued for k_clip, which fills the slot in ----
    if (clip_mask) { // (the same for the whole wave)
        // one place in the queue per wave and atomic: atomics of many lanes or waves on one address
        // are done one after the other, tens of nanoseconds each
        uint32_t base = 0;
        const int leader = __ffsll((long long)clip_mask) - 1;
        if (lane == leader) base = atomicAdd(&a.counters->clip_count, (uint32_t)__popcll(clip_mask));
        base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
        bool no_clip = false, full = false;
        if (clip) {
            const uint32_t rank = (uint32_t)__popcll(alive & below), slot = wslot0 + rank;
            sr[2 * rank] = make_uint4((uint32_t)kNoRaster, 0u, 0u, 0u);
            sr[2 * rank + 1] = make_uint4(0u, 0u, 0u, 0u);
            ss[3 * rank] = make_uint4(0u, 0u, 0u, 0u);
            ss[3 * rank + 1] = make_uint4(0u, 0u, 0u, 0u);
            ss[3 * rank + 2] = make_uint4(0u, prim, 0u, 0u);
            if (!TEX) sc[3 * rank] = sc[3 * rank + 1] = sc[3 * rank + 2] = make_uint4(0u, 0u, 0u, 0u);
            const uint32_t at = base + (uint32_t)__popcll(clip_mask & below);
            if (a.clip_off) {
                no_clip = true;
            } else if (at < a.fan_capacity) {
                ClipItem it;
                it.inst = bd.inst; it.tri = t; it.slot = slot; it.view = view;
                a.clip_list[at] = it;
            } else {
                full = true;
            }
        }
        const uint32_t flags = (__ballot(no_clip) ? kOverflowNoClip : 0u) | (__ballot(full) ? kOverflowClip : 0u);
        if (flags && lane == leader) atomicOr(&a.counters->overflow, flags);
    }
    __builtin_amdgcn_wave_barrier();
    // the wave's staged records to its slots (its own LDS writes are done when it reads them back)
    {
        uint4 *dst = reinterpret_cast<uint4 *>(a.rast + wslot0);
#pragma unroll
        for (uint32_t k = 0; k < 2u; ++k) {
            const uint32_t c = (uint32_t)lane + k * 64u;
            if (c < 2u * n_alive) stream_store(dst + c, sr[c]);
        }
        dst = reinterpret_cast<uint4 *>(a.shade + wslot0);
#pragma unroll
        for (uint32_t k = 0; k < 3u; ++k) {
            const uint32_t c = (uint32_t)lane + k * 64u;
            if (c < 3u * n_alive) stream_store(dst + c, ss[c]);
        }
        if (!TEX) {
            dst = reinterpret_cast<uint4 *>(a.color + wslot0);
#pragma unroll
            for (uint32_t k = 0; k < 3u; ++k) {
                const uint32_t c = (uint32_t)lane + k * 64u;
                if (c < 3u * n_alive) stream_store(dst + c, sc[c]);
            }
        }
    }
}

// (58 registers and 20 KB of LDS: eight workgroups per compute unit; the kernel runs the same with
// seven and 11 % slower with six)
// DEV: built with the development switches of SetupArgs::debug; the ordinary build has none of
// their tests in its code.
// TEX: texture shading only (the usual case), so the colour planes, the loads for them and their
// share of the staging area drop out.
template <bool DEV, bool TEX>
__global__ __launch_bounds__(kSetupThreads) void k_setup(SetupArgs a)
{
    if (!DEV) a.debug = 0u;
    if (TEX) a.shading = SXE_SHADE_TEXTURE;
    // one work item per workgroup (several in a loop measured slower, and a loop keeps per-thread
    // addresses alive across it, which this kernel has no registers for)
    __shared__ SetupShared<TEX> sh;
    setup_block<TEX>(a, blockIdx.x, blockIdx.y, sh); // grid: x over the work items of a view, y over the views
}

// One thread per queued triangle: Sutherland-Hodgman, then the fan of the clipped polygon goes to
// the extra record region; the triangle's own slot only ranks it and points at the fan.
__global__ __launch_bounds__(64) void k_clip(SetupArgs a)
{
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n = min(a.counters->clip_count, a.fan_capacity);
    if (idx >= n) return;
    const ClipItem it = a.clip_list[idx];
    const InstDev inst = a.insts[it.inst];
    const MeshDev mesh = a.meshes[inst.mesh];
    const float *pvm = a.pvm + 16 * ((size_t)it.view * a.n_objects + inst.object);
    const uint32_t *ip = a.indices + mesh.idx_base + 3 * (size_t)it.tri;
    const uint32_t slot = it.slot, prim = inst.prim_base + it.tri;
    ClipVert poly[10], tmp[10];
    for (int k = 0; k < 3; ++k) {
        const sxe_vertex &w = a.vertices[mesh.vtx_base + ip[k]];
        const float4 c = transform_vertex(pvm, w.pos[0], w.pos[1], w.pos[2]);
        poly[k].c[0] = c.x; poly[k].c[1] = c.y; poly[k].c[2] = c.z; poly[k].c[3] = c.w;
        poly[k].c[4] = w.uv[0]; poly[k].c[5] = w.uv[1];
        poly[k].c[6] = w.color[0]; poly[k].c[7] = w.color[1]; poly[k].c[8] = w.color[2];
    }
    float2 lit = make_float2(0.0f, 0.0f); // (of the whole triangle: its pieces lie in the same plane)
    if (a.shading == SXE_SHADE_LIT)
        lit = light_terms(a.models + 16 * (size_t)inst.object, a.vertices[mesh.vtx_base + ip[0]].pos, a.vertices[mesh.vtx_base + ip[1]].pos,
                          a.vertices[mesh.vtx_base + ip[2]].pos, a.light);
    const int n_poly = clip_polygon(poly, tmp, 3);
    const uint32_t n_fan = n_poly >= 3 ? (uint32_t)(n_poly - 2) : 0u;
    const uint32_t fan0 = n_fan ? atomicAdd(&a.counters->fan_records, n_fan) : 0u;
    const bool room = fan0 + n_fan <= a.fan_capacity;
    ShadeRec h;
    h.iw0 = h.diw1 = h.diw2 = h.uw0 = h.duw1 = h.duw2 = h.vw0 = h.dvw1 = h.dvw2 = 0.0f;
    h.prim = prim;
    const uint32_t fan_base = a.g.views * a.n_blocks * kSetupThreads + fan0;
    h.fan_or_ra = fan_base;
    h.tex_fan = ((uint32_t)inst.texture & 0xFFFFFFu) | ((room ? n_fan : 0u) << 24);
    a.shade[slot] = h;
    if (!room) {
        atomicOr(&a.counters->overflow, kOverflowClip);
        return;
    }
    RastRec none;
    none.x0 = kNoRaster; none.y0 = 0; none.d1 = 0; none.d2 = 0; none.z0 = none.dz1 = none.dz2 = none.rA = 0.0f;
    for (uint32_t f = 0; f < n_fan; ++f) {
        const ClipVert &q0 = poly[0], &q1 = poly[f + 1], &q2 = poly[f + 2];
        const uint32_t rec = fan_base + f;
        a.fan_order[fan0 + f] = slot; // the fan ranks where its primitive does
        ScreenTri s2;
        const PVert v0 = project_vertex(make_float4(q0.c[0], q0.c[1], q0.c[2], q0.c[3]), a.g);
        const PVert v1 = project_vertex(make_float4(q1.c[0], q1.c[1], q1.c[2], q1.c[3]), a.g);
        const PVert v2 = project_vertex(make_float4(q2.c[0], q2.c[1], q2.c[2], q2.c[3]), a.g);
        if (!setup_screen_tri(v0, v1, v2, a.g, s2)) {
            a.rast[rec] = none;
            continue;
        }
        Attr3 u = {q0.c[4] * v0.rw, q1.c[4] * v1.rw, q2.c[4] * v2.rw}, v = {q0.c[5] * v0.rw, q1.c[5] * v1.rw, q2.c[5] * v2.rw};
        Attr3 cr = {q0.c[6], q1.c[6], q2.c[6]}, cg = {q0.c[7], q1.c[7], q2.c[7]}, cb = {q0.c[8], q1.c[8], q2.c[8]};
        write_records(a, rec, s2, prim, inst.texture, u, v, cr, cg, cb, lit);
        count_box(a, it.view, s2, translucent_texture(a, inst.texture));
        a.walk_list[atomicAdd(&a.counters->walk_count, 1u)] = rec; // (clipping is rare: one atomic per fan triangle)
    }
}

// ------------------------------------------------------------------------------------------
// Finish: totals, launch order, late bin entries
// ------------------------------------------------------------------------------------------

__device__ inline uint32_t wave_inclusive_scan(uint32_t v, int lane)
{
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(v, d);
        if (lane >= d) v += y;
    }
    return v;
}

constexpr int kOrderClasses = 128;
__device__ inline uint32_t order_class(uint32_t count) { return min((count + 15u) >> 4, (uint32_t)kOrderClasses - 1u); }

// Expected running time of a tile's workgroup, in units of one small-box entry: measured tile
// durations fit 1 : 1 : 2 : 7 for the four size classes and 100 for a late entry.  The largest
// class counts double that: the tiles that finish last are those with a few dozen entries of it
// and nothing else (triangles that cover the tile), and with 16 the tile kernel ends 2 % earlier
// than with 8 (12 and 24 lie in between).
__device__ inline uint32_t tile_cost(const TileCtr &t)
{
    return (t.cnt[0] + t.cnt[1] + 2u * t.cnt[2] + 16u * t.cnt[3] + 100u * t.multi) >> 2;
}

// Tiles are ordered in chunks, a few per thread, one workgroup per chunk.
constexpr uint32_t kFinishThreads = 512;
constexpr int kOrderPerThread = 2; // (8: 4 us more for this kernel on a 4K frame; 1: the tile kernel's tail grows)
constexpr uint32_t kOrderChunk = kFinishThreads * kOrderPerThread;

// counter[cls] += 1 for every lane with `on`; returns the value before the lane's own increment.
// Lanes that share the class of the first such lane (runs of empty or alike tiles are common) go
// together in one atomic: LDS atomics of one instruction on one address are done one by one.
__device__ inline uint32_t class_take(uint32_t *counter, uint32_t cls, bool on, int lane)
{
    const unsigned long long active = __ballot(on);
    if (!active) return 0;
    const int leader = __builtin_amdgcn_readfirstlane(__ffsll((long long)active) - 1);
    const uint32_t lead_cls = (uint32_t)__builtin_amdgcn_readlane((int)cls, leader);
    const bool grouped = on && cls == lead_cls;
    const unsigned long long group = __ballot(grouped);
    uint32_t at = 0;
    if (lane == leader) at = atomicAdd(&counter[cls], (uint32_t)__popcll(group));
    at = (uint32_t)__builtin_amdgcn_readlane((int)at, leader) + (uint32_t)__popcll(group & ((1ull << lane) - 1ull));
    if (on && !grouped) at = atomicAdd(&counter[cls], 1u);
    return at;
}

// Block of tiles a triangle's pixel box touches, from its raster record.
__device__ inline bool block_from_record(const FinishArgs &a, uint32_t rec, uint32_t &cols, uint32_t &rows, uint32_t &first)
{
    const RastRec r = a.rast[rec];
    int32_t bx0 = 0, bx1 = 0, by0 = 0, by1 = 0;
    if (r.x0 == kNoRaster ||
        !pixel_box(unpack_tri(r, a.wide, rec), 0, (int32_t)a.g.width - 1, (int32_t)a.g.sy0, (int32_t)a.g.sy1 - 1, bx0, bx1, by0, by1))
        return false;
    cols = (uint32_t)(bx1 / kTile - bx0 / kTile) + 1u;
    rows = (uint32_t)(by1 / kTile - by0 / kTile) + 1u;
    // the view comes from the slot that ranks the record (its own, or its primitive's for a fan)
    const uint32_t slots_per_view = a.n_blocks * kSetupThreads, n_slots = a.g.views * slots_per_view;
    const uint32_t view = (rec < n_slots ? rec : a.fan_order[rec - n_slots]) / slots_per_view;
    first = view * a.g.n_tiles + (uint32_t)(by0 / kTile - (int32_t)a.g.tile_row0) * a.g.tiles_x + (uint32_t)(bx0 / kTile);
    return true;
}

// Triangles over more than four tiles (and fan triangles) take one entry in the late region of
// each bin they touch.  All (triangle, tile) pairs of the wave are dealt out over its lanes, 64
// per round, so a round costs one chain of memory round trips however many triangles are involved.
// Every lane of the wave must call this.
__device__ inline void bin_walk(const FinishArgs &a, bool walk, uint32_t rec, uint32_t cols, uint32_t rows, uint32_t first)
{
    const unsigned long long todo = __ballot(walk);
    if (!todo) return;
    const int lane = (int)(threadIdx.x & 63u);
    const uint32_t n_items = walk ? cols * rows : 0u;
    uint32_t total = 0;
    for (unsigned long long m = todo; m; m &= m - 1)
        total += (uint32_t)__builtin_amdgcn_readlane((int)n_items, __builtin_amdgcn_readfirstlane(__ffsll((long long)m) - 1));
    const uint32_t cap = a.bl.cap[kSizeClasses], off = a.bl.off[kSizeClasses];
    bool over = false;
    for (uint32_t base = 0; base < total; base += 64u) {
        const uint32_t i = base + (uint32_t)lane;
        uint32_t owner = 0, local = 0, acc = 0;
        for (unsigned long long m = todo; m; m &= m - 1) {
            const int src = __builtin_amdgcn_readfirstlane(__ffsll((long long)m) - 1);
            const uint32_t n_w = (uint32_t)__builtin_amdgcn_readlane((int)n_items, src);
            if (i >= acc && i - acc < n_w) {
                owner = (uint32_t)src;
                local = i - acc;
            }
            acc += n_w;
        }
        const uint32_t w_cols = (uint32_t)__shfl((int)cols, (int)owner), w_first = (uint32_t)__shfl((int)first, (int)owner);
        const uint32_t w_rec = (uint32_t)__shfl((int)rec, (int)owner);
        if (i < total) {
            const uint32_t cy = local / w_cols, cx = local - cy * w_cols;
            const uint32_t tl = w_first + cy * a.g.tiles_x + cx;
            const uint32_t at = atomicAdd(&a.late_fill[tl], 1u);
            if (at < cap)
                a.bins[tl * a.bl.stride + off + at] = w_rec;
            else
                over = true;
        }
    }
    if (__ballot(over) && lane == 0) atomicOr(&a.counters->overflow, kOverflowBins);
}

// What a tile's counters add to the frame's totals.
struct TileTotals {
    unsigned long long pairs;
    uint32_t need[kBinRegions];
    uint32_t blend_max; // most entries of any tile that takes the ordered path
};
__device__ inline void account(TileTotals &tt, const TileCtr &t, bool on)
{
    if (!on) return;
    tt.pairs += (unsigned long long)bin_reserved(t) + t.multi;
    if (t.flags & kTileBlend) tt.blend_max = max(tt.blend_max, bin_size(t));
#pragma unroll
    for (int r = 0; r < kSizeClasses; ++r) tt.need[r] = max(tt.need[r], t.cnt[r]);
    tt.need[kSizeClasses] = max(tt.need[kSizeClasses], t.multi);
}

// Grid (C = chunks of 4096 tiles):
//   workgroups 0..C-1:     launch order of a chunk's tiles and their launch records, and the chunk's
//                          share of the frame's totals: bin entries, the fullest region of each kind
//                          (what the bins would need), capacity check;
//   workgroups C..C+7:     a share of the survivor count each;
//   workgroups C+8..:      the late entries, the listed triangles dealt out over all waves, so that
//                          the dependent memory round trips of a crowd of them (a mesh close to the
//                          camera) are spread out instead of queueing up in one wave.
constexpr uint32_t kWalkChunk = 16;
constexpr uint32_t kCountBlocks = 8;
__global__ __launch_bounds__(kFinishThreads) void k_finish(FinishArgs a)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t n_tiles = total_tiles(a.g);       // of all views
    const uint32_t last_tile = max(n_tiles, 1u) - 1u; // (the counter array never holds fewer than one tile)
    const uint32_t chunks = max((n_tiles + kOrderChunk - 1u) / kOrderChunk, 1u);
    const uint32_t n_blocks = a.n_blocks * a.g.views * (kSetupThreads / 64); // (one count per wave of a setup work item)
    if (blockIdx.x < chunks) {
        // Launch order of the tile workgroups: dearest first by estimated cost, so that the
        // long-running tiles start early and the short ones fill the end of the launch.  Chunk c
        // takes tiles c, c + C, c + 2C, ...: every chunk is a sample of the whole frame, orders its
        // tiles by itself (a counting sort), and the chunks' orders are dealt into the launch
        // order like hands of cards -- no workgroup needs to know another's tiles.
        const uint32_t chunk = blockIdx.x;
        const uint32_t n_mine = n_tiles > chunk ? (n_tiles - chunk + chunks - 1u) / chunks : 0u; // tiles of this chunk
        __shared__ uint32_t s_hist[kOrderClasses];
        if (tid < kOrderClasses) s_hist[tid] = 0;
        __syncthreads();
        TileTotals tt;
        tt.pairs = 0;
        tt.blend_max = 0;
#pragma unroll
        for (int r = 0; r < kBinRegions; ++r) tt.need[r] = 0;
        TileCtr mine[kOrderPerThread]; // the counters are read once
#pragma unroll
        for (int k = 0; k < kOrderPerThread; ++k)
            mine[k] = a.tiles[min(((uint32_t)tid + kFinishThreads * (uint32_t)k) * chunks + chunk, last_tile)];
#pragma unroll
        for (int k = 0; k < kOrderPerThread; ++k) {
            const bool on = (uint32_t)tid + kFinishThreads * (uint32_t)k < n_mine;
            class_take(s_hist, order_class(tile_cost(mine[k])), on, lane);
            account(tt, mine[k], on);
        }
        __syncthreads();
        if (wave == 0) {
            // exclusive scan from the dearest class down: lane l holds classes 127 - 2l and 126 - 2l
            static_assert(kOrderClasses == 128, "two classes per lane of one wave");
            const uint32_t h0 = s_hist[kOrderClasses - 1 - 2 * lane], h1 = s_hist[kOrderClasses - 2 - 2 * lane];
            const uint32_t before = wave_inclusive_scan(h0 + h1, lane) - (h0 + h1);
            s_hist[kOrderClasses - 1 - 2 * lane] = before;
            s_hist[kOrderClasses - 2 - 2 * lane] = before + h0;
        }
        __syncthreads();
        // The chunks' r-th tiles follow one another; the first `full` ranks exist in every chunk.
        const uint32_t full = n_tiles / chunks; // (the first n_tiles - full * chunks chunks have one tile more)
#pragma unroll
        for (int k = 0; k < kOrderPerThread; ++k) {
            const uint32_t j = (uint32_t)tid + kFinishThreads * (uint32_t)k;
            const uint32_t r = class_take(s_hist, order_class(tile_cost(mine[k])), j < n_mine, lane);
            if (j < n_mine) {
                const uint32_t i = j * chunks + chunk; // the tile
                const uint32_t at = r < full ? r * chunks + chunk : full * chunks + chunk; // (rank `full`: only the longer chunks)
                uint4 *dst = reinterpret_cast<uint4 *>(a.tile_launch + at);
                const uint32_t view = i / a.g.n_tiles;
                dst[0] = make_uint4(i - view * a.g.n_tiles, mine[k].multi, mine[k].cnt[3], mine[k].cnt[2]);
                dst[1] = make_uint4(mine[k].cnt[1], mine[k].cnt[0], view, mine[k].flags);
            }
        }
        // The frame's totals: every chunk adds its share (one pass through LDS for all seven
        // figures; lanes 0..6 of the first wave then own one each).  The pair total is kept in
        // 64 bits: a frame inside the stated limits can hold more than 2^32 (triangle, tile) pairs
        // (a few ten thousand screen-filling triangles on an 8K frame).
        __shared__ unsigned long long s_red[kFinishThreads / 64][8];
        {
            unsigned long long v = tt.pairs;
            for (int d = 32; d; d >>= 1) v += __shfl_xor(v, d);
            uint32_t m[6] = {tt.need[0], tt.need[1], tt.need[2], tt.need[3], tt.need[4], tt.blend_max};
#pragma unroll
            for (int q = 0; q < 6; ++q)
                for (int d = 32; d; d >>= 1) m[q] = max(m[q], (uint32_t)__shfl_xor((int)m[q], d));
            if (lane == 0) {
                s_red[wave][0] = v;
#pragma unroll
                for (int q = 0; q < 6; ++q) s_red[wave][1 + q] = m[q];
            }
        }
        __syncthreads();
        if (wave == 0) {
            static_assert(kBinRegions == 5, "figures 1..5 are the regions, 6 the ordered path's fullest tile");
            unsigned long long mine_v = 0;
            if (lane < 7)
                for (int w = 0; w < (int)kFinishThreads / 64; ++w)
                    mine_v = lane == 0 ? mine_v + s_red[w][0] : max(mine_v, s_red[w][lane]);
            // (a chunk's own figures tell whether a region is too small: the frame's are their maxima)
            uint32_t flags = 0;
            if (lane == 0) atomicAdd(&a.counters->total_pairs, mine_v);
            if (lane >= 1 && lane <= 5) {
                atomicMax(&a.counters->need[lane - 1], (uint32_t)mine_v);
                if ((uint32_t)mine_v > pick_region(a.bl.cap, (uint32_t)(lane - 1))) flags = kOverflowBins;
            }
            if (lane == 6) {
                atomicMax(&a.counters->blend_max, (uint32_t)mine_v); // (statistics: k_raster_blend takes such a tile in pieces)
            }
            for (int d = 32; d; d >>= 1) flags |= (uint32_t)__shfl_xor((int)flags, d);
            if (lane == 0 && flags) atomicOr(&a.counters->overflow, flags);
        }
        return;
    }
    // Late entries.  A wave takes up to sixteen listed triangles per round; when the list is short,
    // fewer, so that all waves have work (a triangle's entries are one chain of memory round trips
    // per 64 tiles it touches, and chains in one wave run one after the other).
    if (blockIdx.x < chunks + kCountBlocks) {
        // a share of the survivor counts (statistics): four counts per load, sixteen loads in flight
        // per thread, one atomic per workgroup (atomics on one address are done one after the other)
        __shared__ uint32_t s_sum[kFinishThreads / 64];
        const uint32_t cb = blockIdx.x - chunks;
        const uint32_t n4 = n_blocks / 4u, from = (uint32_t)((unsigned long long)n4 * cb / kCountBlocks),
                       to = (uint32_t)((unsigned long long)n4 * (cb + 1u) / kCountBlocks);
        const uint4 *c4 = reinterpret_cast<const uint4 *>(a.blk_count);
        uint32_t survivors = 0;
        for (uint32_t i = from + (uint32_t)tid; i < to; i += 16u * kFinishThreads) {
            uint4 v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = (i + kFinishThreads * k < to) ? c4[i + kFinishThreads * k] : make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int k = 0; k < 16; ++k) survivors += (v[k].x + v[k].y) + (v[k].z + v[k].w);
        }
        if (cb == 0 && (uint32_t)tid < n_blocks - 4u * n4) survivors += a.blk_count[4u * n4 + tid];
        for (int d = 32; d; d >>= 1) survivors += (uint32_t)__shfl_xor((int)survivors, d);
        if (lane == 0) s_sum[wave] = survivors;
        __syncthreads();
        if (tid == 0) {
            uint32_t all = 0;
            for (int w = 0; w < (int)kFinishThreads / 64; ++w) all += s_sum[w];
            if (all) atomicAdd(&a.counters->total_records, all); // (at most 256 per work item, 2^23 work items)
        }
        return;
    }
    const uint32_t wb = blockIdx.x - chunks - kCountBlocks;
    const uint32_t n_walk = a.counters->walk_count;
    constexpr uint32_t kWaves = kFinishThreads / 64;
    const uint32_t n_waves = a.walk_blocks * kWaves;
    const uint32_t per_wave = min(max((n_walk + n_waves - 1u) / n_waves, 1u), kWalkChunk);
    const uint32_t wave_id = wb * kWaves + (uint32_t)wave;
    for (uint32_t base = wave_id * per_wave; base < n_walk; base += n_waves * per_wave) {
        const bool mine = (uint32_t)lane < per_wave && base + (uint32_t)lane < n_walk;
        uint32_t rec = 0, cols = 0, rows = 0, first = 0;
        bool walk = false;
        if (mine) {
            rec = a.walk_list[base + (uint32_t)lane];
            walk = block_from_record(a, rec, cols, rows, first);
        }
        bin_walk(a, walk, rec, cols, rows, first);
    }
}

// ------------------------------------------------------------------------------------------
// Launchers
// ------------------------------------------------------------------------------------------

hipError_t launch_object(const ObjectArgs &a, hipStream_t s)
{
    // (one wave per workgroup: the objects spread over four times as many compute units as with 256
    // threads, and a kernel that is one chain of dependent steps per thread ends half a microsecond earlier)
    hipLaunchKernelGGL(k_object, dim3(a.n / 64 + 1, a.g.views ? a.g.views : 1u), dim3(64), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_transform_to_model(const sxe_transform *t, float *models, uint32_t n, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_transform_to_model, dim3((n + 255) / 256), dim3(256), 0, s, t, models, n);
    return hipGetLastError();
}

hipError_t launch_rotation_system(sxe_transform *t, float *models, uint32_t n, uint64_t seed, uint32_t frame,
                                  uint32_t n_frames, float delta_ms, hipStream_t s)
{
    if (n == 0 || n_frames == 0) return hipSuccess;
    hipLaunchKernelGGL(k_rotation_system, dim3((n + 255) / 256), dim3(256), 0, s, t, models, n, seed, frame, n_frames, delta_ms);
    return hipGetLastError();
}

hipError_t launch_vertex(const float *px, const float *py, const float *pz, const MeshDev *meshes,
                         const VBlockDesc *vblocks, uint32_t n_vblocks, const float *pvm, float4 *clip, hipStream_t s)
{
    if (n_vblocks == 0) return hipSuccess;
    hipLaunchKernelGGL(k_vertex, dim3(n_vblocks), dim3(256), 0, s, px, py, pz, meshes, vblocks, pvm, clip);
    return hipGetLastError();
}

hipError_t launch_setup(const SetupArgs &a, hipStream_t s)
{
    if (a.n_blocks == 0) return hipSuccess;
    const dim3 grid(a.n_blocks, a.g.views), block(kSetupThreads);
    const bool tex = a.shading == SXE_SHADE_TEXTURE;
    if (a.debug) {
        if (tex) hipLaunchKernelGGL((k_setup<true, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((k_setup<true, false>), grid, block, 0, s, a);
    } else {
        if (tex) hipLaunchKernelGGL((k_setup<false, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((k_setup<false, false>), grid, block, 0, s, a);
    }
    return hipGetLastError();
}

hipError_t launch_clip(const SetupArgs &a, hipStream_t s)
{
    // sized for a full clip list; threads beyond the device-side count return at once
    if (a.n_blocks == 0 || a.fan_capacity == 0) return hipSuccess;
    hipLaunchKernelGGL(k_clip, dim3((a.fan_capacity + 63) / 64), dim3(64), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_finish(const FinishArgs &a, hipStream_t s)
{
    const uint32_t n_tiles = total_tiles(a.g);
    const uint32_t chunks = n_tiles > kOrderChunk ? (n_tiles + kOrderChunk - 1u) / kOrderChunk : 1u;
    hipLaunchKernelGGL(k_finish, dim3(chunks + kCountBlocks + a.walk_blocks), dim3(kFinishThreads), 0, s, a);
    return hipGetLastError();
}

} // namespace sxe
