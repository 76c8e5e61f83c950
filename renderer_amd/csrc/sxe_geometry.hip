// sxe_geometry.hip -- gfx950 kernels of the geometry half of the raster/shade path.
//
// Frame pipeline (all on one stream, no host round trip):
//   k_object        PVM = (P*V)*M per object (standard.vert:28, matrix part) and a conservative
//                   whole-object visibility test
//   k_setup         one thread per submitted triangle: vertex transform, trivial reject, perspective
//                   divide, snap, orientation, bounding box, exact coverage test for tiny boxes;
//                   writes one record per surviving triangle in submission order and counts bin
//                   entries per screen tile
//   k_clip          triangles that cross the near, far or a guard-band plane: Sutherland-Hodgman
//                   and one record per triangle of the resulting fan
//   k_scan          exclusive scan of the tile counts, totals, capacity check
//   k_bin           scatters record slots into the per-tile bins (small boxes to the front of a
//                   bin, the others to its back)
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

// PVM per object, plus: is the whole object certainly invisible?  The eight corners of the mesh's
// bounding box are transformed like vertices; if all of them lie outside one plane of the view
// volume by more than a bound on the rounding error of such a transform, every vertex of the mesh
// does too, so every one of its triangles would be rejected one by one anyway.  The same holds
// for objects that project wholly above or below the scissor rows.  Skipping such objects changes
// no pixel.
__global__ void k_object(ObjectArgs a)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    Mat4 M;
    const float4 *src = reinterpret_cast<const float4 *>(a.models + 16 * (size_t)i);
    for (int c = 0; c < 4; ++c) {
        const float4 v = src[c];
        M.m[4 * c] = v.x; M.m[4 * c + 1] = v.y; M.m[4 * c + 2] = v.z; M.m[4 * c + 3] = v.w;
    }
    const Mat4 R = mul(a.pv, M);
    float4 *dst = reinterpret_cast<float4 *>(a.pvm + 16 * (size_t)i);
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
    a.visible[i] = vis ? 1 : 0;
}

__device__ inline void store_model(float *models, uint32_t i, const Mat4 &R)
{
    float4 *dst = reinterpret_cast<float4 *>(models + 16 * (size_t)i);
    for (int c = 0; c < 4; ++c) dst[c] = make_float4(R.m[4 * c], R.m[4 * c + 1], R.m[4 * c + 2], R.m[4 * c + 3]);
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

// Rotation::run (src/systems/rotation.rs:26-37) followed by to_ubo.
__global__ void k_rotation_system(sxe_transform *__restrict__ t, float *__restrict__ models, uint32_t n,
                                  uint64_t seed, uint32_t frame, float delta_ms)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    sxe_transform tr = t[i];
    const float delta_time = delta_ms * 0.01f;
    const float r = rand01(seed, frame, i);
    const float ud = r * delta_time * 2.0f;
    const Quat by = quat_from_euler_deg(2.0f * ud, 10.0f * ud, 20.0f * ud);
    const Quat q0 = {tr.rotation[0], tr.rotation[1], tr.rotation[2], tr.rotation[3]};
    const Quat q = mul(q0, by);
    tr.rotation[0] = q.s; tr.rotation[1] = q.x; tr.rotation[2] = q.y; tr.rotation[3] = q.z;
    t[i] = tr;
    store_model(models, i, model_matrix(tr.position, tr.scale, q));
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
    bool swapped;               // corners 1 and 2 were exchanged
};

// Perspective divide, viewport map, snap, orientation, bounding box.  False = nothing to draw.
__device__ inline bool setup_screen_tri(const float4 &c0, const float4 &c1, const float4 &c2, const FrameGeom &g,
                                        ScreenTri &s)
{
    if (!(c0.w > 0.0f) || !(c1.w > 0.0f) || !(c2.w > 0.0f)) return false;
    const float rw0 = 1.0f / c0.w, rw1 = 1.0f / c1.w, rw2 = 1.0f / c2.w;
    const float xs0 = ((c0.x * rw0 + 1.0f) * g.half_w) * 256.0f, ys0 = ((c0.y * rw0 + 1.0f) * g.half_h) * 256.0f;
    const float xs1 = ((c1.x * rw1 + 1.0f) * g.half_w) * 256.0f, ys1 = ((c1.y * rw1 + 1.0f) * g.half_h) * 256.0f;
    const float xs2 = ((c2.x * rw2 + 1.0f) * g.half_w) * 256.0f, ys2 = ((c2.y * rw2 + 1.0f) * g.half_h) * 256.0f;
    const float lim = 1073741824.0f;
    if (!(fabsf(xs0) <= lim) || !(fabsf(ys0) <= lim) || !(fabsf(xs1) <= lim) || !(fabsf(ys1) <= lim) ||
        !(fabsf(xs2) <= lim) || !(fabsf(ys2) <= lim))
        return false;
    Tri t;
    t.x0 = (int32_t)rintf(xs0); t.y0 = (int32_t)rintf(ys0);
    t.x1 = (int32_t)rintf(xs1); t.y1 = (int32_t)rintf(ys1);
    t.x2 = (int32_t)rintf(xs2); t.y2 = (int32_t)rintf(ys2);
    const long long area2 = (long long)(t.x1 - t.x0) * (long long)(t.y2 - t.y0) - (long long)(t.x2 - t.x0) * (long long)(t.y1 - t.y0);
    if (area2 == 0) return false;
    float zn1 = c1.z * rw1, zn2 = c2.z * rw2, r1 = rw1, r2 = rw2;
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
    s.zn0 = c0.z * rw0; s.zn1 = zn1; s.zn2 = zn2;
    s.rw0 = rw0; s.rw1 = r1; s.rw2 = r2;
    return true;
}

struct Attr3 {
    float a0, a1, a2;
};

// Writes the records of one screen triangle into slot `rec`.  uv / rgb are per-corner attributes
// in the original corner order; corners 1 and 2 are exchanged when the triangle was reoriented.
__device__ inline void write_records(const SetupArgs &a, uint32_t rec, uint32_t order, const ScreenTri &s, uint32_t prim,
                                     int32_t texture, Attr3 u, Attr3 v, Attr3 cr, Attr3 cg, Attr3 cb)
{
    RastRec r;
    r.x0 = s.t.x0; r.y0 = s.t.y0;
    const int32_t dx1 = s.t.x1 - s.t.x0, dy1 = s.t.y1 - s.t.y0, dx2 = s.t.x2 - s.t.x0, dy2 = s.t.y2 - s.t.y0;
    if (fits16(dx1) && fits16(dy1) && fits16(dx2) && fits16(dy2)) {
        r.d1 = ((uint32_t)dx1 & 0xFFFFu) | ((uint32_t)dy1 << 16);
        r.d2 = ((uint32_t)dx2 & 0xFFFFu) | ((uint32_t)dy2 << 16);
    } else {
        r.d1 = kWideTri;
        r.d2 = 0;
        WideRec w;
        w.x1 = s.t.x1; w.y1 = s.t.y1; w.x2 = s.t.x2; w.y2 = s.t.y2;
        a.wide[rec] = w;
    }
    r.z0 = s.zn0; r.dz1 = s.zn1 - s.zn0; r.dz2 = s.zn2 - s.zn0;
    r.order = order;
    a.rast[rec] = r;
    if (s.swapped) {
        float t;
        t = u.a1; u.a1 = u.a2; u.a2 = t;
        t = v.a1; v.a1 = v.a2; v.a2 = t;
        t = cr.a1; cr.a1 = cr.a2; cr.a2 = t;
        t = cg.a1; cg.a1 = cg.a2; cg.a2 = t;
        t = cb.a1; cb.a1 = cb.a2; cb.a2 = t;
    }
    ShadeRec h;
    h.iw0 = s.rw0; h.diw1 = s.rw1 - s.rw0; h.diw2 = s.rw2 - s.rw0;
    if (a.shading != SXE_SHADE_VERTEX_COLOR) {
        const float uw0 = u.a0 * s.rw0, uw1 = u.a1 * s.rw1, uw2 = u.a2 * s.rw2;
        const float vw0 = v.a0 * s.rw0, vw1 = v.a1 * s.rw1, vw2 = v.a2 * s.rw2;
        h.uw0 = uw0; h.duw1 = uw1 - uw0; h.duw2 = uw2 - uw0;
        h.vw0 = vw0; h.dvw1 = vw1 - vw0; h.dvw2 = vw2 - vw0;
    } else {
        h.uw0 = h.duw1 = h.duw2 = h.vw0 = h.dvw1 = h.dvw2 = 0.0f;
    }
    h.prim = prim;
    h.fan_base = 0;
    h.tex_fan = (uint32_t)texture & 0xFFFFFFu;
    a.shade[rec] = h;
    if (a.shading != SXE_SHADE_TEXTURE) {
        const float r0 = cr.a0 * s.rw0, r1 = cr.a1 * s.rw1, r2 = cr.a2 * s.rw2;
        const float g0 = cg.a0 * s.rw0, g1 = cg.a1 * s.rw1, g2 = cg.a2 * s.rw2;
        const float b0 = cb.a0 * s.rw0, b1 = cb.a1 * s.rw1, b2 = cb.a2 * s.rw2;
        ColorRec c;
        c.rw0 = r0; c.drw1 = r1 - r0; c.drw2 = r2 - r0;
        c.gw0 = g0; c.dgw1 = g1 - g0; c.dgw2 = g2 - g0;
        c.bw0 = b0; c.dbw1 = b1 - b0; c.dbw2 = b2 - b0;
        c.pad0 = c.pad1 = c.pad2 = 0.0f;
        a.color[rec] = c;
    }
}

// One atomic per tile of the box: used for triangles that span several tiles and for fan triangles.
__device__ inline void count_box(const SetupArgs &a, const ScreenTri &s)
{
    const int32_t tx0 = s.px0 / kTile, tx1 = s.px1 / kTile;
    const int32_t ty0 = s.py0 / kTile - (int32_t)a.g.tile_row0, ty1 = s.py1 / kTile - (int32_t)a.g.tile_row0;
    for (int32_t ty = ty0; ty <= ty1; ++ty)
        for (int32_t tx = tx0; tx <= tx1; ++tx) atomicAdd(&a.tile_count[(uint32_t)ty * a.g.tiles_x + (uint32_t)tx], 1u);
}

// One thread per submitted triangle; a workgroup covers 256 consecutive triangles of one object
// and owns 256 consecutive record slots, which it fills from the front in submission order.  That
// order is what makes "first primitive wins a depth tie" (Comparison::Less,
// src/renderer/drawer.rs:700) fall out of a (depth | slot) minimum.
__global__ __launch_bounds__(kSetupThreads) void k_setup(SetupArgs a)
{
    __shared__ uint32_t s_wave_tot[kSetupThreads / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const BlockDesc bd = a.blocks[blockIdx.x];
    const InstDev inst = a.insts[bd.inst];
    if (!a.visible[inst.object]) {
        if (tid == 0) a.blk_count[blockIdx.x] = 0;
        return;
    }
    const MeshDev mesh = a.meshes[inst.mesh];
    const float *pvm = a.pvm + 16 * (size_t)inst.object;
    const uint32_t t = bd.tri_start + tid;
    const bool valid = t < mesh.n_tri;

    enum { CULLED = 0, FAST = 1, CLIP = 2 };
    int state = CULLED;
    uint32_t vi0 = 0, vi1 = 0, vi2 = 0;
    ScreenTri st;
    st.px0 = st.px1 = st.py0 = st.py1 = 0;

    if (valid) {
        const uint32_t *ip = a.indices + mesh.idx_base + 3 * (size_t)t;
        vi0 = mesh.vtx_base + ip[0];
        vi1 = mesh.vtx_base + ip[1];
        vi2 = mesh.vtx_base + ip[2];
        const float *p0 = a.vertices[vi0].pos, *p1 = a.vertices[vi1].pos, *p2 = a.vertices[vi2].pos;
        const float4 c0 = transform_vertex(pvm, p0[0], p0[1], p0[2]);
        const float4 c1 = transform_vertex(pvm, p1[0], p1[1], p1[2]);
        const float4 c2 = transform_vertex(pvm, p2[0], p2[1], p2[2]);
        // all three corners outside one plane of the view volume -> nothing visible
        const bool rej =
            (!(c0.z >= 0.0f) && !(c1.z >= 0.0f) && !(c2.z >= 0.0f)) ||
            (!(c0.w - c0.z >= 0.0f) && !(c1.w - c1.z >= 0.0f) && !(c2.w - c2.z >= 0.0f)) ||
            (!(c0.w + c0.x >= 0.0f) && !(c1.w + c1.x >= 0.0f) && !(c2.w + c2.x >= 0.0f)) ||
            (!(c0.w - c0.x >= 0.0f) && !(c1.w - c1.x >= 0.0f) && !(c2.w - c2.x >= 0.0f)) ||
            (!(c0.w + c0.y >= 0.0f) && !(c1.w + c1.y >= 0.0f) && !(c2.w + c2.y >= 0.0f)) ||
            (!(c0.w - c0.y >= 0.0f) && !(c1.w - c1.y >= 0.0f) && !(c2.w - c2.y >= 0.0f));
        if (!rej) {
            bool inside = true;
            const float4 cs[3] = {c0, c1, c2};
            for (int k = 0; k < 3; ++k) {
                const float gw = kGuard * cs[k].w;
                inside = inside && (cs[k].z >= 0.0f) && (cs[k].w - cs[k].z >= 0.0f) && (gw + cs[k].x >= 0.0f) &&
                         (gw - cs[k].x >= 0.0f) && (gw + cs[k].y >= 0.0f) && (gw - cs[k].y >= 0.0f);
            }
            if (inside) {
                if (setup_screen_tri(c0, c1, c2, a.g, st)) {
                    state = FAST;
                    // a triangle whose box holds a few pixels and contains none of their samples draws nothing
                    const int32_t bw = st.px1 - st.px0 + 1, bh = st.py1 - st.py0 + 1;
                    if (bw * bh <= 4 && narrow(st.t) && !covers_any(st.t, st.px0, st.px1, st.py0, st.py1)) state = CULLED;
                }
            } else {
                state = CLIP; // crosses the near, far or a guard-band plane: k_clip cuts it
            }
        }
    }

    // rank of every survivor inside the workgroup
    const unsigned long long alive = __ballot(state != CULLED);
    const uint32_t wave_rank = (uint32_t)__popcll(alive & ((1ull << lane) - 1ull));
    if (lane == 0) s_wave_tot[wave] = (uint32_t)__popcll(alive);
    __syncthreads();
    uint32_t wave_base = 0, total = 0;
    for (int w = 0; w < kSetupThreads / 64; ++w) {
        const uint32_t v = s_wave_tot[w];
        if (w < wave) wave_base += v;
        total += v;
    }
    if (tid == 0) a.blk_count[blockIdx.x] = total;
    const uint32_t slot = blockIdx.x * kSetupThreads + wave_base + wave_rank;
    const uint32_t prim = inst.prim_base + t;

    // ---- triangles that needed no clipping ----
    if (state == FAST) {
        const sxe_vertex &w0 = a.vertices[vi0], &w1 = a.vertices[vi1], &w2 = a.vertices[vi2];
        Attr3 u = {w0.uv[0], w1.uv[0], w2.uv[0]}, v = {w0.uv[1], w1.uv[1], w2.uv[1]};
        Attr3 cr = {0, 0, 0}, cg = {0, 0, 0}, cb = {0, 0, 0};
        if (a.shading != SXE_SHADE_TEXTURE) {
            cr = {w0.color[0], w1.color[0], w2.color[0]};
            cg = {w0.color[1], w1.color[1], w2.color[1]};
            cb = {w0.color[2], w1.color[2], w2.color[2]};
        }
        write_records(a, slot, slot, st, prim, inst.texture, u, v, cr, cg, cb);
    }
    const int32_t tx0 = st.px0 / kTile, ty0 = st.py0 / kTile;
    const bool single = (state == FAST) && tx0 == st.px1 / kTile && ty0 == st.py1 / kTile;
    const uint32_t tile = single ? (uint32_t)(ty0 - (int32_t)a.g.tile_row0) * a.g.tiles_x + (uint32_t)tx0 : 0xFFFFFFFFu;
    // wave-aggregated tile counting: one atomic per distinct tile per wave
    unsigned long long pending = __ballot(single);
    while (pending) {
        const int leader = __builtin_amdgcn_readfirstlane(__ffsll((long long)pending) - 1);
        const uint32_t lt = __builtin_amdgcn_readlane(tile, leader);
        const unsigned long long same = __ballot(single && tile == lt);
        if (lane == leader) atomicAdd(&a.tile_count[lt], (uint32_t)__popcll(same));
        pending &= ~same;
    }
    if (state == FAST && !single) count_box(a, st);

    // ---- triangles cut by the near/far/guard planes keep their slot (it ranks them in submission
    // order) and are queued for k_clip ----
    if (state == CLIP) {
        RastRec none;
        none.x0 = kNoRaster; none.y0 = 0; none.d1 = 0; none.d2 = 0; none.z0 = none.dz1 = none.dz2 = 0.0f; none.order = slot;
        a.rast[slot] = none;
        const uint32_t at = atomicAdd(&a.counters->clip_count, 1u);
        if (at < a.fan_capacity) {
            ClipItem it;
            it.inst = bd.inst; it.tri = t; it.slot = slot;
            a.clip_list[at] = it;
        } else {
            atomicOr(&a.counters->overflow, kOverflowClip);
        }
    }
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
    const float *pvm = a.pvm + 16 * (size_t)inst.object;
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
    const int n_poly = clip_polygon(poly, tmp, 3);
    const uint32_t n_fan = n_poly >= 3 ? (uint32_t)(n_poly - 2) : 0u;
    const uint32_t fan0 = n_fan ? atomicAdd(&a.counters->fan_records, n_fan) : 0u;
    const bool room = fan0 + n_fan <= a.fan_capacity;
    ShadeRec h;
    h.iw0 = h.diw1 = h.diw2 = h.uw0 = h.duw1 = h.duw2 = h.vw0 = h.dvw1 = h.dvw2 = 0.0f;
    h.prim = prim;
    h.fan_base = a.n_blocks * kSetupThreads + fan0;
    h.tex_fan = ((uint32_t)inst.texture & 0xFFFFFFu) | ((room ? n_fan : 0u) << 24);
    a.shade[slot] = h;
    if (!room) {
        atomicOr(&a.counters->overflow, kOverflowClip);
        return;
    }
    RastRec none;
    none.x0 = kNoRaster; none.y0 = 0; none.d1 = 0; none.d2 = 0; none.z0 = none.dz1 = none.dz2 = 0.0f; none.order = slot;
    for (uint32_t f = 0; f < n_fan; ++f) {
        const ClipVert &q0 = poly[0], &q1 = poly[f + 1], &q2 = poly[f + 2];
        const uint32_t rec = h.fan_base + f;
        ScreenTri s2;
        if (!setup_screen_tri(make_float4(q0.c[0], q0.c[1], q0.c[2], q0.c[3]), make_float4(q1.c[0], q1.c[1], q1.c[2], q1.c[3]),
                              make_float4(q2.c[0], q2.c[1], q2.c[2], q2.c[3]), a.g, s2)) {
            a.rast[rec] = none;
            continue;
        }
        Attr3 u = {q0.c[4], q1.c[4], q2.c[4]}, v = {q0.c[5], q1.c[5], q2.c[5]};
        Attr3 cr = {q0.c[6], q1.c[6], q2.c[6]}, cg = {q0.c[7], q1.c[7], q2.c[7]}, cb = {q0.c[8], q1.c[8], q2.c[8]};
        write_records(a, rec, slot, s2, prim, inst.texture, u, v, cr, cg, cb);
        count_box(a, s2);
    }
}

// ------------------------------------------------------------------------------------------
// Scan
// ------------------------------------------------------------------------------------------

__device__ inline uint32_t wave_inclusive_scan(uint32_t v, int lane)
{
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(v, d);
        if (lane >= d) v += y;
    }
    return v;
}

// Workgroup 0: exclusive scan of the per-tile bin counts (eight counts per thread and round).
// Workgroup 1: total number of surviving triangles.
__global__ __launch_bounds__(1024) void k_scan(ScanArgs a)
{
    __shared__ uint32_t s_part[16];
    __shared__ uint32_t s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (blockIdx.x == 1) {
        uint32_t sum = 0;
        for (uint32_t i = tid; i < a.n_blocks; i += 1024) sum += a.blk_count[i];
        for (int d = 32; d; d >>= 1) sum += __shfl_xor(sum, d);
        if (lane == 0) s_part[wave] = sum;
        __syncthreads();
        if (tid == 0) {
            uint32_t tot = 0;
            for (int w = 0; w < 16; ++w) tot += s_part[w];
            a.counters->total_records = tot;
        }
        return;
    }
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < a.n_tiles; base += 8192) {
        const uint32_t i0 = base + 8u * (uint32_t)tid;
        uint32_t v[8], sum = 0;
        for (int k = 0; k < 8; ++k) {
            v[k] = (i0 + k < a.n_tiles) ? a.tile_count[i0 + k] : 0u;
            sum += v[k];
        }
        const uint32_t incl = wave_inclusive_scan(sum, lane);
        if (lane == 63) s_part[wave] = incl;
        __syncthreads();
        uint32_t wbase = 0, tot = 0;
        for (int w = 0; w < 16; ++w) {
            const uint32_t p = s_part[w];
            if (w < wave) wbase += p;
            tot += p;
        }
        const uint32_t carry = s_carry;
        uint32_t run = carry + wbase + incl - sum;
        for (int k = 0; k < 8; ++k) {
            if (i0 + k < a.n_tiles) a.tile_off[i0 + k] = run;
            run += v[k];
        }
        __syncthreads();
        if (tid == 0) s_carry = carry + tot;
        __syncthreads();
    }
    if (tid == 0) {
        const uint32_t total = s_carry;
        a.counters->total_pairs = total;
        if (total > a.bin_capacity) atomicOr(&a.counters->overflow, kOverflowBins);
    }
}

// ------------------------------------------------------------------------------------------
// Binning
// ------------------------------------------------------------------------------------------

// Reads the surviving records back and scatters their slot numbers into the tile bins.  Entries
// whose pixel box inside the tile holds at most kSmallBox pixels are stored from the front of the
// tile's range, the others from its back, so that the raster kernel meets boxes of similar size
// next to each other.  The order of entries inside a bin is irrelevant to the image.
__global__ __launch_bounds__(kSetupThreads) void k_bin(BinArgs a)
{
    if (a.counters->overflow) return;
    const int tid = threadIdx.x, lane = tid & 63;
    uint32_t rec;
    bool valid;
    if (blockIdx.x < a.n_blocks) {
        const uint32_t cnt = a.blk_count[blockIdx.x];
        if (cnt == 0) return;
        valid = (uint32_t)tid < cnt;
        rec = blockIdx.x * kSetupThreads + tid;
    } else {
        const uint32_t first = (blockIdx.x - a.n_blocks) * kSetupThreads;
        const uint32_t n_fan = a.counters->fan_records;
        if (first >= n_fan) return;
        valid = first + tid < n_fan;
        rec = a.n_blocks * kSetupThreads + first + tid;
    }
    Tri t = {0, 0, 0, 0, 0, 0};
    int32_t bx0 = 0, bx1 = 0, by0 = 0, by1 = 0;
    if (valid) {
        const RastRec r = a.rast[rec];
        valid = r.x0 != kNoRaster;
        if (valid) {
            t = unpack_tri(r, a.wide, rec);
            valid = pixel_box(t, 0, (int32_t)a.g.width - 1, (int32_t)a.g.sy0, (int32_t)a.g.sy1 - 1, bx0, bx1, by0, by1);
        }
    }
    const int32_t tx0 = bx0 / kTile, tx1 = bx1 / kTile, ty0 = by0 / kTile, ty1 = by1 / kTile;
    const bool single = valid && tx0 == tx1 && ty0 == ty1;
    const bool big = (bx1 - bx0 + 1) * (by1 - by0 + 1) > kSmallBox;
    const uint32_t tile = (uint32_t)(ty0 - (int32_t)a.g.tile_row0) * a.g.tiles_x + (uint32_t)tx0;
    const uint32_t key = single ? (tile * 2u + (big ? 1u : 0u)) : 0xFFFFFFFFu;
    // group the lanes of the wave by (tile, class) without touching memory ...
    unsigned long long pending = __ballot(single);
    int my_leader = lane;
    uint32_t my_rank = 0, my_count = 0;
    while (pending) {
        const int leader = __builtin_amdgcn_readfirstlane(__ffsll((long long)pending) - 1);
        const uint32_t lk = __builtin_amdgcn_readlane(key, leader);
        const bool mine = single && key == lk;
        const unsigned long long same = __ballot(mine);
        if (mine) {
            my_leader = leader;
            my_rank = (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
            my_count = (uint32_t)__popcll(same);
        }
        pending &= ~same;
    }
    // ... then every group leader reserves its run of bin entries in one round of atomics
    uint32_t base = 0;
    if (single && lane == my_leader) base = atomicAdd((key & 1u) ? &a.fill_back[tile] : &a.fill_front[tile], my_count);
    base = __shfl(base, my_leader);
    if (single) {
        const uint32_t p = base + my_rank;
        const uint32_t off = a.tile_off[tile];
        a.bins[(key & 1u) ? off + a.tile_count[tile] - 1u - p : off + p] = rec;
    }
    if (valid && !single) {
        for (int32_t ty = ty0; ty <= ty1; ++ty)
            for (int32_t tx = tx0; tx <= tx1; ++tx) {
                const uint32_t tl = (uint32_t)(ty - (int32_t)a.g.tile_row0) * a.g.tiles_x + (uint32_t)tx;
                const int32_t w = min(bx1, tx * kTile + kTile - 1) - max(bx0, tx * kTile) + 1;
                const int32_t h = min(by1, ty * kTile + kTile - 1) - max(by0, ty * kTile) + 1;
                const uint32_t off = a.tile_off[tl];
                if (w * h > kSmallBox) {
                    const uint32_t p = atomicAdd(&a.fill_back[tl], 1u);
                    a.bins[off + a.tile_count[tl] - 1u - p] = rec;
                } else {
                    const uint32_t p = atomicAdd(&a.fill_front[tl], 1u);
                    a.bins[off + p] = rec;
                }
            }
    }
}

// ------------------------------------------------------------------------------------------
// Launchers
// ------------------------------------------------------------------------------------------

hipError_t launch_object(const ObjectArgs &a, hipStream_t s)
{
    if (a.n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_object, dim3((a.n + 255) / 256), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_transform_to_model(const sxe_transform *t, float *models, uint32_t n, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_transform_to_model, dim3((n + 255) / 256), dim3(256), 0, s, t, models, n);
    return hipGetLastError();
}

hipError_t launch_rotation_system(sxe_transform *t, float *models, uint32_t n, uint64_t seed, uint32_t frame,
                                  float delta_ms, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_rotation_system, dim3((n + 255) / 256), dim3(256), 0, s, t, models, n, seed, frame, delta_ms);
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
    hipLaunchKernelGGL(k_setup, dim3(a.n_blocks), dim3(kSetupThreads), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_clip(const SetupArgs &a, hipStream_t s)
{
    // sized for a full clip list; threads beyond the device-side count return at once
    if (a.n_blocks == 0 || a.fan_capacity == 0) return hipSuccess;
    hipLaunchKernelGGL(k_clip, dim3((a.fan_capacity + 63) / 64), dim3(64), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_scan(const ScanArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(k_scan, dim3(2), dim3(1024), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_bin(const BinArgs &a, hipStream_t s)
{
    if (a.n_blocks + a.fan_blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(k_bin, dim3(a.n_blocks + a.fan_blocks), dim3(kSetupThreads), 0, s, a);
    return hipGetLastError();
}

} // namespace sxe
