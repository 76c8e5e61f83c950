// sxe_kernels.hip -- gfx950 kernels of the raster/shade path.
//
// Frame pipeline (all on one stream, no host round trip):
//   k_object        PVM = (P*V)*M per object                              (standard.vert:28, matrix part)
//   k_setup<false>  per triangle: vertex transform, cull/clip, snap, bbox; counts record slots per
//                   workgroup and bin entries per screen tile
//   k_scan          exclusive scans of both count arrays, capacity check
//   k_setup<true>   same decisions again; writes triangle records in submission order and scatters
//                   their slot numbers into the per-tile bins
//   k_raster_shade  one workgroup per 32x32 tile: coverage + depth in LDS with 64-bit
//                   (depth|slot) minima, then one deferred shade + store per pixel
//
// What the kernels replace in the reference is fixed-function Vulkan state selected at
// src/renderer/drawer.rs:652-705 plus the two shaders under src/data/shaders/.
//
// Numeric contract: every float expression is evaluated as written (built with
// -ffp-contract=off); coverage is exact integer arithmetic on 24.8 fixed-point vertices.
#include "sxe_internal.h"
#include "sxe_srgb_tables.h"

namespace sxe {

__constant__ float c_srgb_decode[256];
__constant__ float c_unorm8[256];
__constant__ float c_srgb_thr[256];
static bool g_tables_ready[64] = {};

static hipError_t ensure_tables()
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 64 && g_tables_ready[dev]) return hipSuccess;
    float thr[256];
    for (int i = 0; i < 255; ++i) thr[i] = SXE_SRGB_ENCODE_THR[i];
    thr[255] = __builtin_inff();
    if ((e = hipMemcpyToSymbol(HIP_SYMBOL(c_srgb_decode), SXE_SRGB_DECODE, sizeof(float) * 256)) != hipSuccess) return e;
    if ((e = hipMemcpyToSymbol(HIP_SYMBOL(c_unorm8), SXE_UNORM8, sizeof(float) * 256)) != hipSuccess) return e;
    if ((e = hipMemcpyToSymbol(HIP_SYMBOL(c_srgb_thr), thr, sizeof(float) * 256)) != hipSuccess) return e;
    if (dev < 64) g_tables_ready[dev] = true;
    return hipSuccess;
}

// ------------------------------------------------------------------------------------------
// Per-object kernels
// ------------------------------------------------------------------------------------------

__global__ void k_object(const float *__restrict__ models, Mat4 pv, float *__restrict__ pvm, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Mat4 M;
    const float4 *src = reinterpret_cast<const float4 *>(models + 16 * (size_t)i);
    for (int c = 0; c < 4; ++c) {
        const float4 v = src[c];
        M.m[4 * c] = v.x; M.m[4 * c + 1] = v.y; M.m[4 * c + 2] = v.z; M.m[4 * c + 3] = v.w;
    }
    const Mat4 R = mul(pv, M);
    float4 *dst = reinterpret_cast<float4 *>(pvm + 16 * (size_t)i);
    for (int c = 0; c < 4; ++c) dst[c] = make_float4(R.m[4 * c], R.m[4 * c + 1], R.m[4 * c + 2], R.m[4 * c + 3]);
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
    int32_t X0, Y0, X1, Y1, X2, Y2; // 24.8 fixed point, oriented so that the doubled area is > 0
    int32_t px0, px1, py0, py1;     // inclusive pixel bounding box, clamped to the scissor
    float zn0, zn1, zn2;            // z/w per corner (after reorientation)
    float rw0, rw1, rw2;            // 1/w per corner
    float rA;                       // 1 / doubled area
    bool swapped;                   // corners 1 and 2 were exchanged
};

// Perspective divide, viewport map, snap, orientation, bounding box.  False = nothing to draw.
__device__ inline bool setup_screen_tri(const float4 &c0, const float4 &c1, const float4 &c2,
                                        const FrameGeom &g, ScreenTri &s)
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
    const int32_t X0 = (int32_t)rintf(xs0), Y0 = (int32_t)rintf(ys0);
    int32_t X1 = (int32_t)rintf(xs1), Y1 = (int32_t)rintf(ys1);
    int32_t X2 = (int32_t)rintf(xs2), Y2 = (int32_t)rintf(ys2);
    long long area2 = (long long)(X1 - X0) * (long long)(Y2 - Y0) - (long long)(X2 - X0) * (long long)(Y1 - Y0);
    if (area2 == 0) return false;
    float zn1 = c1.z * rw1, zn2 = c2.z * rw2, r1 = rw1, r2 = rw2;
    s.swapped = area2 < 0;
    if (s.swapped) {
        int32_t t;
        t = X1; X1 = X2; X2 = t;
        t = Y1; Y1 = Y2; Y2 = t;
        float f;
        f = zn1; zn1 = zn2; zn2 = f;
        f = r1; r1 = r2; r2 = f;
        area2 = -area2;
    }
    const int32_t xmin = min(X0, min(X1, X2)), xmax = max(X0, max(X1, X2));
    const int32_t ymin = min(Y0, min(Y1, Y2)), ymax = max(Y0, max(Y1, Y2));
    // pixel (i,j) is sampled at (256 i + 128, 256 j + 128)
    int32_t px0 = (xmin + 127) >> 8, px1 = (xmax - 128) >> 8;
    int32_t py0 = (ymin + 127) >> 8, py1 = (ymax - 128) >> 8;
    px0 = max(px0, 0);
    px1 = min(px1, (int32_t)g.width - 1);
    py0 = max(py0, (int32_t)g.sy0);
    py1 = min(py1, (int32_t)g.sy1 - 1);
    if (px0 > px1 || py0 > py1) return false;
    s.X0 = X0; s.Y0 = Y0; s.X1 = X1; s.Y1 = Y1; s.X2 = X2; s.Y2 = Y2;
    s.px0 = px0; s.px1 = px1; s.py0 = py0; s.py1 = py1;
    s.zn0 = c0.z * rw0; s.zn1 = zn1; s.zn2 = zn2;
    s.rw0 = rw0; s.rw1 = r1; s.rw2 = r2;
    s.rA = 1.0f / (float)area2;
    return true;
}

__device__ inline uint32_t wave_inclusive_scan(uint32_t v, int lane)
{
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(v, d);
        if (lane >= d) v += y;
    }
    return v;
}

struct Attr3 {
    float a0, a1, a2;
};

// Writes the record of one screen triangle.  uv / rgb are per-corner attributes in the original
// corner order; corner 1 and 2 are exchanged when the triangle was reoriented.
__device__ inline void write_records(const SetupArgs &a, uint32_t slot, const ScreenTri &s, uint32_t prim,
                                     int32_t texture, Attr3 u, Attr3 v, Attr3 cr, Attr3 cg, Attr3 cb)
{
    RastRec r;
    r.x0 = s.X0; r.y0 = s.Y0; r.x1 = s.X1; r.y1 = s.Y1; r.x2 = s.X2; r.y2 = s.Y2;
    r.z0 = s.zn0; r.dz1 = s.zn1 - s.zn0; r.dz2 = s.zn2 - s.zn0; r.rA = s.rA;
    r.prim = prim; r.texture = texture;
    a.rast[slot] = r;
    if (s.swapped) {
        float t;
        t = u.a1; u.a1 = u.a2; u.a2 = t;
        t = v.a1; v.a1 = v.a2; v.a2 = t;
        t = cr.a1; cr.a1 = cr.a2; cr.a2 = t;
        t = cg.a1; cg.a1 = cg.a2; cg.a2 = t;
        t = cb.a1; cb.a1 = cb.a2; cb.a2 = t;
    }
    if (a.shading != SXE_SHADE_VERTEX_COLOR) {
        const float uw0 = u.a0 * s.rw0, uw1 = u.a1 * s.rw1, uw2 = u.a2 * s.rw2;
        const float vw0 = v.a0 * s.rw0, vw1 = v.a1 * s.rw1, vw2 = v.a2 * s.rw2;
        ShadeRec h;
        h.iw0 = s.rw0; h.diw1 = s.rw1 - s.rw0; h.diw2 = s.rw2 - s.rw0;
        h.uw0 = uw0; h.duw1 = uw1 - uw0; h.duw2 = uw2 - uw0;
        h.vw0 = vw0; h.dvw1 = vw1 - vw0; h.dvw2 = vw2 - vw0;
        h.pad0 = h.pad1 = h.pad2 = 0.0f;
        a.shade[slot] = h;
    } else {
        ShadeRec h;
        h.iw0 = s.rw0; h.diw1 = s.rw1 - s.rw0; h.diw2 = s.rw2 - s.rw0;
        h.uw0 = h.duw1 = h.duw2 = h.vw0 = h.dvw1 = h.dvw2 = 0.0f;
        h.pad0 = h.pad1 = h.pad2 = 0.0f;
        a.shade[slot] = h;
    }
    if (a.shading != SXE_SHADE_TEXTURE) {
        const float r0 = cr.a0 * s.rw0, r1 = cr.a1 * s.rw1, r2 = cr.a2 * s.rw2;
        const float g0 = cg.a0 * s.rw0, g1 = cg.a1 * s.rw1, g2 = cg.a2 * s.rw2;
        const float b0 = cb.a0 * s.rw0, b1 = cb.a1 * s.rw1, b2 = cb.a2 * s.rw2;
        ColorRec c;
        c.rw0 = r0; c.drw1 = r1 - r0; c.drw2 = r2 - r0;
        c.gw0 = g0; c.dgw1 = g1 - g0; c.dgw2 = g2 - g0;
        c.bw0 = b0; c.dbw1 = b1 - b0; c.dbw2 = b2 - b0;
        c.pad0 = c.pad1 = c.pad2 = 0.0f;
        a.color[slot] = c;
    }
}

// Bin a triangle whose pixel box spans several tiles (or comes from the clipper): one atomic per tile.
template <bool EMIT>
__device__ inline void bin_box(const SetupArgs &a, const ScreenTri &s, uint32_t slot)
{
    const int32_t tx0 = s.px0 / kTile, tx1 = s.px1 / kTile;
    const int32_t ty0 = s.py0 / kTile - (int32_t)a.g.tile_row0, ty1 = s.py1 / kTile - (int32_t)a.g.tile_row0;
    for (int32_t ty = ty0; ty <= ty1; ++ty)
        for (int32_t tx = tx0; tx <= tx1; ++tx) {
            const uint32_t tile = (uint32_t)ty * a.g.tiles_x + (uint32_t)tx;
            if (EMIT) {
                const uint32_t pos = atomicAdd(&a.tile_fill[tile], 1u);
                a.bins[a.tile_off[tile] + pos] = slot;
            } else {
                atomicAdd(&a.tile_count[tile], 1u);
            }
        }
}

// One thread per submitted triangle; a workgroup covers 256 consecutive triangles of one object,
// so record slots (workgroup base + rank inside the workgroup) follow submission order, which is
// what makes "first primitive wins a depth tie" fall out of a (depth|slot) minimum.
template <bool EMIT>
__global__ __launch_bounds__(kSetupThreads) void k_setup(SetupArgs a)
{
    __shared__ uint32_t s_wave_tot[kSetupThreads / 64];
    if (EMIT && a.counters->overflow) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const BlockDesc bd = a.blocks[blockIdx.x];
    const InstDev inst = a.insts[bd.inst];
    const MeshDev mesh = a.meshes[inst.mesh];
    const float *pvm = a.pvm + 16 * (size_t)inst.object;
    const uint32_t t = bd.tri_start + tid;
    const bool valid = t < mesh.n_tri;

    enum { CULLED = 0, FAST = 1, CLIP = 2 };
    int state = CULLED;
    uint32_t vi0 = 0, vi1 = 0, vi2 = 0;
    float4 c0 = make_float4(0, 0, 0, 0), c1 = c0, c2 = c0;
    ScreenTri st;
    st.px0 = st.px1 = st.py0 = st.py1 = 0;
    uint32_t n_out = 0;

    if (valid) {
        const uint32_t *ip = a.indices + mesh.idx_base + 3 * (size_t)t;
        vi0 = mesh.vtx_base + ip[0];
        vi1 = mesh.vtx_base + ip[1];
        vi2 = mesh.vtx_base + ip[2];
        const float *p0 = a.vertices[vi0].pos, *p1 = a.vertices[vi1].pos, *p2 = a.vertices[vi2].pos;
        c0 = transform_vertex(pvm, p0[0], p0[1], p0[2]);
        c1 = transform_vertex(pvm, p1[0], p1[1], p1[2]);
        c2 = transform_vertex(pvm, p2[0], p2[1], p2[2]);
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
                    n_out = 1;
                }
            } else {
                // reserve one slot per fan triangle of the clipped polygon
                ClipVert poly[10], tmp[10];
                const float4 cs2[3] = {c0, c1, c2};
                for (int k = 0; k < 3; ++k) {
                    poly[k].c[0] = cs2[k].x; poly[k].c[1] = cs2[k].y; poly[k].c[2] = cs2[k].z; poly[k].c[3] = cs2[k].w;
                    for (int q = 4; q < 9; ++q) poly[k].c[q] = 0.0f;
                }
                const int n = clip_polygon(poly, tmp, 3);
                if (n >= 3) {
                    state = CLIP;
                    n_out = (uint32_t)(n - 2);
                }
            }
        }
    }

    // workgroup exclusive scan of n_out
    const uint32_t incl = wave_inclusive_scan(n_out, lane);
    if (lane == 63) s_wave_tot[wave] = incl;
    __syncthreads();
    uint32_t wave_base = 0, total = 0;
    for (int w = 0; w < kSetupThreads / 64; ++w) {
        const uint32_t v = s_wave_tot[w];
        if (w < wave) wave_base += v;
        total += v;
    }
    const uint32_t rank = wave_base + incl - n_out;
    uint32_t slot = rank;
    if (EMIT) {
        slot += a.blk_base[blockIdx.x];
    } else if (tid == 0) {
        a.blk_count[blockIdx.x] = total;
    }

    // ---- triangles that needed no clipping ----
    if (EMIT && state == FAST) {
        const sxe_vertex &w0 = a.vertices[vi0], &w1 = a.vertices[vi1], &w2 = a.vertices[vi2];
        Attr3 u = {w0.uv[0], w1.uv[0], w2.uv[0]}, v = {w0.uv[1], w1.uv[1], w2.uv[1]};
        Attr3 cr = {0, 0, 0}, cg = {0, 0, 0}, cb = {0, 0, 0};
        if (a.shading != SXE_SHADE_TEXTURE) {
            cr = {w0.color[0], w1.color[0], w2.color[0]};
            cg = {w0.color[1], w1.color[1], w2.color[1]};
            cb = {w0.color[2], w1.color[2], w2.color[2]};
        }
        write_records(a, slot, st, inst.prim_base + t, inst.texture, u, v, cr, cg, cb);
    }
    const int32_t tx0 = st.px0 / kTile, ty0 = st.py0 / kTile;
    const bool single = (state == FAST) && tx0 == st.px1 / kTile && ty0 == st.py1 / kTile;
    const uint32_t tile = single ? (uint32_t)(ty0 - (int32_t)a.g.tile_row0) * a.g.tiles_x + (uint32_t)tx0 : 0xFFFFFFFFu;
    // wave-aggregated bin allocation: one atomic per distinct tile per wave
    unsigned long long pending = __ballot(single);
    while (pending) {
        const int leader = __ffsll((long long)pending) - 1;
        const uint32_t lt = __shfl(tile, leader);
        const bool mine = single && tile == lt;
        const unsigned long long same = __ballot(mine);
        uint32_t base = 0;
        if (lane == leader) {
            const uint32_t cnt = (uint32_t)__popcll(same);
            base = EMIT ? atomicAdd(&a.tile_fill[lt], cnt) : atomicAdd(&a.tile_count[lt], cnt);
        }
        if (EMIT) {
            base = __shfl(base, leader);
            if (mine) {
                const uint32_t r = (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
                a.bins[a.tile_off[lt] + base + r] = slot;
            }
        }
        pending &= ~same;
    }
    if (state == FAST && !single) bin_box<EMIT>(a, st, slot);

    // ---- triangles cut by the near/far/guard planes: fan of the clipped polygon ----
    if (state == CLIP) {
        ClipVert poly[10], tmp[10];
        const float4 cs[3] = {c0, c1, c2};
        const uint32_t vis[3] = {vi0, vi1, vi2};
        for (int k = 0; k < 3; ++k) {
            const sxe_vertex &w = a.vertices[vis[k]];
            poly[k].c[0] = cs[k].x; poly[k].c[1] = cs[k].y; poly[k].c[2] = cs[k].z; poly[k].c[3] = cs[k].w;
            poly[k].c[4] = w.uv[0]; poly[k].c[5] = w.uv[1];
            poly[k].c[6] = w.color[0]; poly[k].c[7] = w.color[1]; poly[k].c[8] = w.color[2];
        }
        const int n = clip_polygon(poly, tmp, 3);
        for (int f = 1; f + 1 < n; ++f) {
            const ClipVert &q0 = poly[0], &q1 = poly[f], &q2 = poly[f + 1];
            ScreenTri s2;
            if (!setup_screen_tri(make_float4(q0.c[0], q0.c[1], q0.c[2], q0.c[3]),
                                  make_float4(q1.c[0], q1.c[1], q1.c[2], q1.c[3]),
                                  make_float4(q2.c[0], q2.c[1], q2.c[2], q2.c[3]), a.g, s2))
                continue;
            const uint32_t sl = slot + (uint32_t)(f - 1);
            if (EMIT) {
                Attr3 u = {q0.c[4], q1.c[4], q2.c[4]}, v = {q0.c[5], q1.c[5], q2.c[5]};
                Attr3 cr = {q0.c[6], q1.c[6], q2.c[6]}, cg = {q0.c[7], q1.c[7], q2.c[7]}, cb = {q0.c[8], q1.c[8], q2.c[8]};
                write_records(a, sl, s2, inst.prim_base + t, inst.texture, u, v, cr, cg, cb);
            }
            bin_box<EMIT>(a, s2, sl);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Scans
// ------------------------------------------------------------------------------------------

// Workgroup 0 scans the per-workgroup record counts, workgroup 1 the per-tile bin counts.
__global__ __launch_bounds__(1024) void k_scan(ScanArgs a)
{
    __shared__ uint32_t s_part[16];
    __shared__ uint32_t s_carry;
    const bool tiles = blockIdx.x == 1;
    const uint32_t *src = tiles ? a.tile_count : a.blk_count;
    uint32_t *dst = tiles ? a.tile_off : a.blk_base;
    const uint32_t n = tiles ? a.n_tiles : a.n_blocks;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += 1024) {
        const uint32_t i = base + tid;
        const uint32_t v = i < n ? src[i] : 0u;
        const uint32_t incl = wave_inclusive_scan(v, lane);
        if (lane == 63) s_part[wave] = incl;
        __syncthreads();
        uint32_t wbase = 0, tot = 0;
        for (int w = 0; w < 16; ++w) {
            const uint32_t p = s_part[w];
            if (w < wave) wbase += p;
            tot += p;
        }
        const uint32_t carry = s_carry;
        if (i < n) dst[i] = carry + wbase + incl - v;
        __syncthreads();
        if (tid == 0) s_carry = carry + tot;
        __syncthreads();
    }
    if (tid == 0) {
        const uint32_t total = s_carry;
        if (tiles) {
            a.counters->total_pairs = total;
            if (total > a.bin_capacity) atomicOr(&a.counters->overflow, 2u);
        } else {
            a.counters->total_records = total;
            if (total > a.rec_capacity) atomicOr(&a.counters->overflow, 1u);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Tile raster + deferred shade
// ------------------------------------------------------------------------------------------

__device__ inline uint32_t srgb_encode_lds(const float *thr, float x)
{
    // number of thresholds <= x; thr[255] = +inf
    uint32_t lo = 0;
    for (uint32_t step = 128; step; step >>= 1)
        if (thr[lo + step - 1] <= x) lo += step;
    return lo;
}

__device__ inline uint32_t unorm8_encode(float a)
{
    if (!(a > 0.0f)) a = 0.0f;
    if (a > 1.0f) a = 1.0f;
    return (uint32_t)(int)rintf(a * 255.0f);
}

// texture(sampler2D, uv) for a single-level RGBA8 sRGB image, linear filter, clamp to edge
// (standard.frag:12; sampler at src/renderer/drawer.rs:231; format at src/renderer/types.rs:186).
__device__ inline void sample_texture(const uint8_t *__restrict__ texels, const TexDev &t, const float *dec,
                                      float u, float v, float out[4])
{
    const float fw = (float)t.width, fh = (float)t.height;
    float x = u * fw - 0.5f, y = v * fh - 0.5f;
    if (!(x >= -1.0f)) x = -1.0f;
    if (x > fw) x = fw;
    if (!(y >= -1.0f)) y = -1.0f;
    if (y > fh) y = fh;
    const float fx0 = floorf(x), fy0 = floorf(y);
    const float ax = x - fx0, ay = y - fy0;
    const int mw = (int)t.width - 1, mh = (int)t.height - 1;
    const int ix0 = min(max((int)fx0, 0), mw), ix1 = min(max((int)fx0 + 1, 0), mw);
    const int iy0 = min(max((int)fy0, 0), mh), iy1 = min(max((int)fy0 + 1, 0), mh);
    const uint32_t *tx = reinterpret_cast<const uint32_t *>(texels) + t.offset;
    const uint32_t t00 = tx[(size_t)iy0 * t.width + ix0], t10 = tx[(size_t)iy0 * t.width + ix1];
    const uint32_t t01 = tx[(size_t)iy1 * t.width + ix0], t11 = tx[(size_t)iy1 * t.width + ix1];
    for (int k = 0; k < 4; ++k) {
        const int sh = 8 * k;
        float c00, c10, c01, c11;
        if (k < 3) {
            c00 = dec[(t00 >> sh) & 255]; c10 = dec[(t10 >> sh) & 255];
            c01 = dec[(t01 >> sh) & 255]; c11 = dec[(t11 >> sh) & 255];
        } else {
            c00 = c_unorm8[(t00 >> sh) & 255]; c10 = c_unorm8[(t10 >> sh) & 255];
            c01 = c_unorm8[(t01 >> sh) & 255]; c11 = c_unorm8[(t11 >> sh) & 255];
        }
        const float top = c00 + ax * (c10 - c00);
        const float bot = c01 + ax * (c11 - c01);
        out[k] = top + ay * (bot - top);
    }
}

__device__ inline float clamp_depth(float z)
{
    if (!(z > 0.0f)) z = 0.0f;
    return z;
}

template <bool COVERAGE>
__global__ __launch_bounds__(kRasterThreads) void k_raster_shade(RasterArgs a)
{
    __shared__ unsigned long long vis[kTilePixels]; // (depth bits << 32) | record slot, per pixel
    __shared__ float s_thr[256];
    __shared__ float s_dec[256];
    if (a.counters->overflow) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t tile = blockIdx.x;
    const int32_t tpx = (int32_t)(tile % a.g.tiles_x) * kTile;
    const int32_t tpy = (int32_t)(tile / a.g.tiles_x + a.g.tile_row0) * kTile;
    for (int i = tid; i < kTilePixels; i += kRasterThreads) vis[i] = kVisEmpty;
    s_thr[tid] = c_srgb_thr[tid];
    s_dec[tid] = c_srgb_decode[tid];
    __syncthreads();

    // rectangle of this tile that exists in the frame and the scissor (inclusive)
    const int32_t cx0 = tpx, cx1 = min(tpx + kTile - 1, (int32_t)a.g.width - 1);
    const int32_t cy0 = max(tpy, (int32_t)a.g.sy0), cy1 = min(tpy + kTile - 1, (int32_t)a.g.sy1 - 1);

    const uint32_t n = (a.debug & kDebugSkipRaster) ? 0u : a.tile_count[tile], off = a.tile_off[tile];
    for (uint32_t base = (uint32_t)wave * 64u; base < n; base += kRasterThreads) {
        const uint32_t e = base + (uint32_t)lane;
        const bool valid = e < n;
        uint32_t slot = 0;
        bool small = false, large = false;
        if (valid) {
            slot = a.bins[off + e];
            const RastRec r = a.rast[slot];
            const int32_t xmin = min(r.x0, min(r.x1, r.x2)), xmax = max(r.x0, max(r.x1, r.x2));
            const int32_t ymin = min(r.y0, min(r.y1, r.y2)), ymax = max(r.y0, max(r.y1, r.y2));
            const int32_t bx0 = max((xmin + 127) >> 8, cx0), bx1 = min((xmax - 128) >> 8, cx1);
            const int32_t by0 = max((ymin + 127) >> 8, cy0), by1 = min((ymax - 128) >> 8, cy1);
            const bool nonempty = bx0 <= bx1 && by0 <= by1;
            const int32_t bw = bx1 - bx0 + 1, bh = by1 - by0 + 1;
            small = nonempty && bw * bh <= 16 && (xmax - xmin) < 16384 && (ymax - ymin) < 16384;
            large = nonempty && !small;
            if (a.debug & kDebugSkipSmall) small = false;
            if (a.debug & kDebugSkipLarge) large = false;
            if (small) {
                // every difference below is < 2^14 and every product < 2^28: exact in 32 bits
                const int32_t ex0 = r.x2 - r.x1, ey0 = r.y2 - r.y1;
                const int32_t ex1 = r.x0 - r.x2, ey1 = r.y0 - r.y2;
                const int32_t ex2 = r.x1 - r.x0, ey2 = r.y1 - r.y0;
                // top-left rule: an edge function of exactly 0 counts only on top or left edges
                const int32_t b0 = (ey0 < 0 || (ey0 == 0 && ex0 > 0)) ? 0 : 1;
                const int32_t b1 = (ey1 < 0 || (ey1 == 0 && ex1 > 0)) ? 0 : 1;
                const int32_t b2 = (ey2 < 0 || (ey2 == 0 && ex2 > 0)) ? 0 : 1;
                int32_t i = bx0, j = by0;
                const int32_t cnt = bw * bh;
                for (int32_t k = 0; k < cnt; ++k) {
                    const int32_t sx = 256 * i + 128, sy = 256 * j + 128;
                    const int32_t E0 = ex0 * (sy - r.y1) - ey0 * (sx - r.x1);
                    const int32_t E1 = ex1 * (sy - r.y2) - ey1 * (sx - r.x2);
                    const int32_t E2 = ex2 * (sy - r.y0) - ey2 * (sx - r.x0);
                    if (((E0 - b0) | (E1 - b1) | (E2 - b2)) >= 0) {
                        const float l1 = (float)E1 * r.rA, l2 = (float)E2 * r.rA;
                        const float z = clamp_depth((r.z0 + l1 * r.dz1) + l2 * r.dz2);
                        if (z < 1.0f) {
                            const unsigned long long key = ((unsigned long long)__float_as_uint(z) << 32) | slot;
                            unsigned long long *p = &vis[(j - tpy) * kTile + (i - tpx)];
                            if (!(a.debug & kDebugSkipAtomic) && key < *p) atomicMin(p, key);
                        }
                    }
                    if (++i > bx1) { i = bx0; ++j; }
                }
            }
        }
        // triangles with a larger footprint: the whole wave walks the box in 8x8 pixel blocks
        unsigned long long lm = __ballot(large);
        while (lm) {
            const int src = __ffsll((long long)lm) - 1;
            lm &= lm - 1;
            const uint32_t sl = __shfl(slot, src);
            const RastRec r = a.rast[sl];
            const int32_t xmin = min(r.x0, min(r.x1, r.x2)), xmax = max(r.x0, max(r.x1, r.x2));
            const int32_t ymin = min(r.y0, min(r.y1, r.y2)), ymax = max(r.y0, max(r.y1, r.y2));
            const int32_t bx0 = max((xmin + 127) >> 8, cx0), bx1 = min((xmax - 128) >> 8, cx1);
            const int32_t by0 = max((ymin + 127) >> 8, cy0), by1 = min((ymax - 128) >> 8, cy1);
            const long long ex0 = r.x2 - r.x1, ey0 = r.y2 - r.y1;
            const long long ex1 = r.x0 - r.x2, ey1 = r.y0 - r.y2;
            const long long ex2 = r.x1 - r.x0, ey2 = r.y1 - r.y0;
            const long long b0 = (ey0 < 0 || (ey0 == 0 && ex0 > 0)) ? 0 : 1;
            const long long b1 = (ey1 < 0 || (ey1 == 0 && ex1 > 0)) ? 0 : 1;
            const long long b2 = (ey2 < 0 || (ey2 == 0 && ex2 > 0)) ? 0 : 1;
            for (int32_t yb = by0; yb <= by1; yb += 8)
                for (int32_t xb = bx0; xb <= bx1; xb += 8) {
                    const int32_t i = xb + (lane & 7), j = yb + (lane >> 3);
                    if (i > bx1 || j > by1) continue;
                    const long long sx = 256 * i + 128, sy = 256 * j + 128;
                    const long long E0 = ex0 * (sy - r.y1) - ey0 * (sx - r.x1);
                    const long long E1 = ex1 * (sy - r.y2) - ey1 * (sx - r.x2);
                    const long long E2 = ex2 * (sy - r.y0) - ey2 * (sx - r.x0);
                    if (((E0 - b0) | (E1 - b1) | (E2 - b2)) >= 0) {
                        const float l1 = (float)E1 * r.rA, l2 = (float)E2 * r.rA;
                        const float z = clamp_depth((r.z0 + l1 * r.dz1) + l2 * r.dz2);
                        if (z < 1.0f) {
                            const unsigned long long key = ((unsigned long long)__float_as_uint(z) << 32) | sl;
                            unsigned long long *p = &vis[(j - tpy) * kTile + (i - tpx)];
                            if (key < *p) atomicMin(p, key);
                        }
                    }
                }
        }
    }
    __syncthreads();
    if (a.debug & kDebugSkipResolve) return;

    // ---- resolve: shade the winning triangle of every pixel once, blend over the cleared target ----
    const float clr_r = s_dec[a.clear_code & 255], clr_g = s_dec[(a.clear_code >> 8) & 255];
    const float clr_b = s_dec[(a.clear_code >> 16) & 255], clr_a = c_unorm8[(a.clear_code >> 24) & 255];
    for (int k = 0; k < kTilePixels / kRasterThreads; ++k) {
        const int lx = tid & (kTile - 1), ly = (tid / kTile) + k * (kRasterThreads / kTile);
        const int32_t i = tpx + lx, j = tpy + ly;
        if (i > cx1 || j < cy0 || j > cy1) continue;
        const unsigned long long key = vis[ly * kTile + lx];
        const size_t p = (size_t)j * a.g.width + (size_t)i;
        uint32_t code = a.clear_code, prim = 0xFFFFFFFFu;
        float depth = 1.0f;
        if (key != kVisEmpty) {
            const uint32_t slot = (uint32_t)key;
            depth = __uint_as_float((uint32_t)(key >> 32));
            const RastRec r = a.rast[slot];
            const ShadeRec h = a.shade[slot];
            prim = r.prim;
            const long long sx = 256 * i + 128, sy = 256 * j + 128;
            const long long E1 = (long long)(r.x0 - r.x2) * (sy - r.y2) - (long long)(r.y0 - r.y2) * (sx - r.x2);
            const long long E2 = (long long)(r.x1 - r.x0) * (sy - r.y0) - (long long)(r.y1 - r.y0) * (sx - r.x0);
            const float l1 = (float)E1 * r.rA, l2 = (float)E2 * r.rA;
            const float iw = (h.iw0 + l1 * h.diw1) + l2 * h.diw2;
            const float riw = 1.0f / iw;
            float src[4];
            if (a.shading == SXE_SHADE_VERTEX_COLOR) {
                const ColorRec c = a.color[slot];
                src[0] = ((c.rw0 + l1 * c.drw1) + l2 * c.drw2) * riw;
                src[1] = ((c.gw0 + l1 * c.dgw1) + l2 * c.dgw2) * riw;
                src[2] = ((c.bw0 + l1 * c.dbw1) + l2 * c.dbw2) * riw;
                src[3] = 1.0f;
            } else {
                const float u = ((h.uw0 + l1 * h.duw1) + l2 * h.duw2) * riw;
                const float v = ((h.vw0 + l1 * h.dvw1) + l2 * h.dvw2) * riw;
                sample_texture(a.texels, a.textures[r.texture], s_dec, u, v, src);
                if (a.shading == SXE_SHADE_COLOR_TIMES_TEXTURE) {
                    const ColorRec c = a.color[slot];
                    src[0] = (((c.rw0 + l1 * c.drw1) + l2 * c.drw2) * riw) * src[0];
                    src[1] = (((c.gw0 + l1 * c.dgw1) + l2 * c.dgw2) * riw) * src[1];
                    src[2] = (((c.bw0 + l1 * c.dbw1) + l2 * c.dbw2) * riw) * src[2];
                }
            }
            // BlendState::ALPHA (src/renderer/drawer.rs:660-663) over the cleared sRGB8 target
            const float sa = src[3], ia = 1.0f - sa;
            const uint32_t cr = srgb_encode_lds(s_thr, src[0] * sa + clr_r * ia);
            const uint32_t cg = srgb_encode_lds(s_thr, src[1] * sa + clr_g * ia);
            const uint32_t cb = srgb_encode_lds(s_thr, src[2] * sa + clr_b * ia);
            const uint32_t ca = unorm8_encode(sa + clr_a * ia);
            code = cr | (cg << 8) | (cb << 16) | (ca << 24);
        }
        reinterpret_cast<uint32_t *>(a.out_color)[p] = code;
        a.out_depth[p] = depth;
        if (COVERAGE) a.out_prim[p] = prim;
    }
}

// ------------------------------------------------------------------------------------------
// Launchers
// ------------------------------------------------------------------------------------------

hipError_t launch_object(const float *models, const Mat4 &pv, float *pvm, uint32_t n, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_object, dim3((n + 255) / 256), dim3(256), 0, s, models, pv, pvm, n);
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

hipError_t launch_setup_count(const SetupArgs &a, hipStream_t s)
{
    if (a.n_blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(k_setup<false>, dim3(a.n_blocks), dim3(kSetupThreads), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_setup_emit(const SetupArgs &a, hipStream_t s)
{
    if (a.n_blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(k_setup<true>, dim3(a.n_blocks), dim3(kSetupThreads), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_scan(const ScanArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(k_scan, dim3(2), dim3(1024), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_raster_shade(const RasterArgs &a, hipStream_t s)
{
    hipError_t e = ensure_tables();
    if (e != hipSuccess) return e;
    if (a.g.n_tiles == 0) return hipSuccess;
    if (a.out_prim)
        hipLaunchKernelGGL(k_raster_shade<true>, dim3(a.g.n_tiles), dim3(kRasterThreads), 0, s, a);
    else
        hipLaunchKernelGGL(k_raster_shade<false>, dim3(a.g.n_tiles), dim3(kRasterThreads), 0, s, a);
    return hipGetLastError();
}

} // namespace sxe
