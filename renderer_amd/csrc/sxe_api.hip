// sxe_api.hip -- host side of libsxe_hip: the context, scene upload, per-frame submission and
// readback behind the C ABI declared in include/sxe.h.
//
// The context plays the part of the reference's GfxDrawer (src/renderer/drawer.rs:33-52): it owns
// the vertex/index buffers, the per-object uniform data, the camera matrices, the texture table
// and the render targets.  There is no CPU rendering path: every entry point that needs the GPU
// fails with SXE_E_DEVICE when HIP cannot provide one.
#include "sxe_internal.h"
#include "sxe_srgb_tables.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

using namespace sxe;

namespace {

thread_local std::string g_create_error;

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0; // elements
    hipError_t reserve(size_t n)
    {
        if (n <= cap) return hipSuccess;
        if (p) {
            hipError_t e = hipFree(p);
            p = nullptr;
            cap = 0;
            if (e != hipSuccess) return e;
        }
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&p), n * sizeof(T));
        if (e == hipSuccess) cap = n; else p = nullptr;
        return e;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

// The images of a context lie one behind the other in one allocation per plane, so that a set of
// views can render into consecutive images with one launch.
struct Image {
    uint8_t *color = nullptr;
    float *depth = nullptr;
    uint32_t *prim = nullptr;   // null unless the context keeps coverage
    uint8_t *h_color = nullptr; // pinned copy handed out by sxe_map_color (allocated on first use)
    bool drawn = false;
};

enum { EV_BEGIN = 0, EV_OBJECT, EV_SETUP, EV_FINISH, EV_RASTER, EV_N };

// Frames are enqueued without waiting for the one before (the reference keeps one frame per
// swapchain image in flight the same way, src/renderer/drawer.rs:416-424).  What has to outlive a
// frame until the host has looked at its result is kept per slot: the per-object matrices the
// frame was drawn with (a redraw after growing a work buffer reuses them, whatever the transforms
// have become since), the counters it reported and its timing events.
constexpr int kFrameSlots = 3;
struct FrameSlot {
    DevBuf<float> pvm;
    DevBuf<uint8_t> visible;
    FrameCounters *h_counters = nullptr; // pinned
    FrameCounters *d_counters = nullptr; // the same block as the device addresses it
    hipEvent_t ev[EV_N] = {};
    hipEvent_t ev_tr[2] = {};
    hipEvent_t done = nullptr;
    DevBuf<float> models_snap;      // SXE_SHADE_LIT: the model matrices the frame was drawn with
    LightDev light = {};            // and the light
    bool pending = false, transform_timed = false;
    bool timed = false; // this frame's kernels are bracketed by events
    uint32_t image = 0; // first image of the launch set
    uint32_t views = 1; // views in the launch set: they render into images image .. image + views - 1
    DevBuf<float> pv_views; // proj * view of every view of a set of several
};

// The device memory one frame works in besides its inputs and its images: per-object matrices,
// triangle records, bin bookkeeping and the bins themselves.  sxe_draw takes it from the context's
// own buffers, sxe_draw_noalloc from the library's static array.
struct WorkBufs {
    float *pvm = nullptr;
    uint8_t *visible = nullptr;
    uint32_t *blk_count = nullptr, *tile_counters = nullptr, *bins = nullptr, *walk_list = nullptr, *fan_order = nullptr;
    TileLaunch *tile_launch = nullptr;
    BinLayout bl = {};
    RastRec *rast = nullptr;
    WideRec *wide = nullptr;
    ClipItem *clip_list = nullptr;
    ShadeRec *shade = nullptr;
    ColorRec *color = nullptr;
    FrameCounters *counters = nullptr;
    size_t fan_capacity = 0;
    bool with_clip = true; // k_clip follows k_setup
    const float *models_lit = nullptr; // the model matrices k_setup lights with (SXE_SHADE_LIT)
    float *models_snap = nullptr;      // where k_object leaves a copy of them for that, or null
};

} // namespace

// Work memory of sxe_draw_noalloc: one array per GPU the library runs on, part of the loaded code
// object, so no call ever allocates it.  In host code the name `scratch` is only a handle for
// hipGetSymbolAddress; kernels and memsets get the per-device address kept in g_scratch_ptr.
constexpr size_t kScratchBytes = SXE_NOALLOC_SCRATCH_BYTES;
static __device__ __align__(256) unsigned char scratch[kScratchBytes];

namespace {

std::mutex g_scratch_lock;
unsigned char *g_scratch_ptr[kMaxDevices] = {}; // device address of `scratch` on each device, set by sxe_create
std::atomic<bool> g_scratch_busy[kMaxDevices];  // a call of sxe_draw_noalloc is using the device's array

// Holds a device's static array for the duration of one sxe_draw_noalloc call.
struct ScratchHold {
    std::atomic<bool> *flag = nullptr;
    bool mine = false;
    explicit ScratchHold(int device) : flag(&g_scratch_busy[device]) { mine = !flag->exchange(true, std::memory_order_acquire); }
    ~ScratchHold() { if (mine) flag->store(false, std::memory_order_release); }
    ScratchHold(const ScratchHold &) = delete;
    ScratchHold &operator=(const ScratchHold &) = delete;
};

// With `device` current: looks up where this device holds its copy of `scratch`.
hipError_t init_scratch(int device)
{
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> hold(g_scratch_lock);
    if (g_scratch_ptr[device]) return hipSuccess;
    unsigned char *ptr = nullptr;
    hipError_t e = hipGetSymbolAddress(reinterpret_cast<void **>(&ptr), HIP_SYMBOL(scratch));
    if (e != hipSuccess) return e;
    if (!ptr) return hipErrorInvalidSymbol;
    g_scratch_ptr[device] = ptr;
    return hipSuccess;
}

} // namespace

struct sxe_ctx {
    sxe_config cfg;
    FrameGeom g;
    hipStream_t stream = nullptr;
    const uint32_t *last_tile_counters = nullptr; // the per-tile counters of the frame enqueued last (diagnostics)
    std::string err;

    // scene
    bool have_scene = false, have_camera = false;
    bool blend_scene = false;       // some texture of the table has a texel with alpha below 255
    int32_t scene_max_texture = -1; // largest texture index the uploaded scene samples
    bool scene_lost_textures = false; // the texture table was replaced by one that lacks some of them
    uint32_t n_objects = 0, n_draws = 0, n_blocks = 0, n_vblocks = 0, n_meshes = 0, n_textures = 0;
    uint64_t n_prims = 0, n_inst_vertices = 0;
    DevBuf<sxe_vertex> vertices;
    DevBuf<float> pos_x, pos_y, pos_z; // SoA copies for the stand-alone vertex kernel
    DevBuf<uint32_t> indices;
    DevBuf<MeshDev> meshes;
    DevBuf<InstDev> insts;
    DevBuf<BlockDesc> blocks;
    DevBuf<VBlockDesc> vblocks;
    DevBuf<float> models;
    DevBuf<sxe_transform> transforms;
    bool have_transforms = false;
    DevBuf<uint8_t> texels;    // the textures as they were given, back to back (source of the footprint tables)
    DevBuf<uint2> footprints;  // (width + 2) x (height + 1) cells per texture, back to back (one spare at the end: a footprint is two cells)
    DevBuf<TexDev> textab;
    Mat4 view, proj, pv;
    LightDev light; // of SXE_SHADE_LIT

    // per-frame work buffers
    DevBuf<uint32_t> blk_count, tile_counters /* a TileCtr per tile | late-fill counter per tile | launch order */, bins, object_mesh, walk_list, fan_order;
    DevBuf<TileLaunch> tile_launch;
    DevBuf<RastRec> rast;
    DevBuf<WideRec> wide;
    DevBuf<ClipItem> clip_list;
    DevBuf<ShadeRec> shade;
    DevBuf<ColorRec> color;
    DevBuf<FrameCounters> counters;
    DevBuf<float4> clip;
    DevBuf<unsigned long long> tile_times;
    size_t fan_capacity = 0; // records past the per-workgroup segments
    BinLayout bl = {};       // capacities of the five bin regions every tile has; the bin array holds n_tiles * bl.stride entries

    std::vector<Image> images;
    DevBuf<uint8_t> color_pool;
    DevBuf<float> depth_pool;
    DevBuf<uint32_t> prim_pool;
    uint32_t view_capacity = 1; // views per launch set the work buffers are sized for
    FrameSlot slots[kFrameSlots];
    uint64_t frames_enqueued = 0, frames_retired = 0; // frame f lives in slot f % kFrameSlots
    sxe_timings timings = {};
    sxe_timings timing_totals = {};
    uint64_t timed_frames = 0;
    uint32_t timing_stride = 1; // events are recorded for every timing_stride-th frame (0: for none)
    // a rotation-system update that has been asked for but not launched: the next frame's first
    // kernel applies it (anything else that touches transforms or matrices launches it first)
    bool rot_pending = false;
    uint64_t rot_seed = 0;
    uint32_t rot_first = 0, rot_n = 0;
    float rot_delta_ms = 0.0f;
    sxe_frame_stats stats = {};
    uint32_t redraws = 0;
    // The clip kernel is launched only while frames have triangles to clip: after kClipQuiet
    // frames without any it is left out, and the first frame that would have needed it reports
    // an overflow and is drawn again (like a frame whose work buffers were too small).
    bool clip_enabled = true;
    uint32_t clip_quiet = 0;
    // Per-frame uploads (model matrices, transforms) go through a small ring of pinned buffers: the
    // caller's memory is only borrowed for the call, and copying it here first lets the transfer run
    // on the stream without the call waiting for it.
    struct Staging {
        void *p = nullptr;
        size_t cap = 0;
        hipEvent_t done = nullptr;
        bool used = false;
    } staging[4];
    uint32_t staging_next = 0;
};

namespace {

int fail(sxe_ctx *c, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_create_error = buf;
    return code;
}

#define SXE_HIP(c, expr)                                                                              \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess) return fail((c), SXE_E_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

uint32_t srgb_encode_host(float x)
{
    uint32_t n = 0;
    for (int k = 0; k < 255; ++k) n += (SXE_SRGB_ENCODE_THR[k] <= x) ? 1u : 0u;
    return n;
}

uint32_t unorm8_host(float a)
{
    if (!(a > 0.0f)) a = 0.0f;
    if (a > 1.0f) a = 1.0f;
    return (uint32_t)(int)rintf(a * 255.0f);
}

FrameGeom make_geom(const sxe_config &cfg)
{
    FrameGeom g;
    g.width = cfg.width;
    g.height = cfg.height;
    g.sy0 = cfg.scissor_y0;
    g.sy1 = cfg.scissor_y1;
    g.tiles_x = (cfg.width + kTile - 1) / kTile;
    g.tile_row0 = g.sy0 / kTile;
    const uint32_t last_row = (g.sy1 > g.sy0) ? (g.sy1 - 1) / kTile : g.tile_row0;
    g.n_tile_rows = (g.sy1 > g.sy0) ? last_row - g.tile_row0 + 1 : 0;
    g.n_tiles = g.tiles_x * g.n_tile_rows;
    g.half_w = (float)cfg.width * 0.5f;
    g.half_h = (float)cfg.height * 0.5f;
    g.views = 1;
    return g;
}

struct HostMesh {
    const sxe_vertex *v;
    uint32_t nv;
    const uint32_t *idx;
    uint32_t ni;
};

struct HostObject {
    uint32_t mesh;
    int32_t texture;
    uint32_t rendered;
    sxe_transform transform;
};

// Record arrays: 256 slots per setup work item, then fan_capacity slots for clipped fans.
hipError_t reserve_records(sxe_ctx *c)
{
    const size_t n = (size_t)c->view_capacity * c->n_blocks * kSetupThreads + c->fan_capacity;
    hipError_t e;
    if ((e = c->rast.reserve(n)) != hipSuccess) return e;
    if ((e = c->wide.reserve(n)) != hipSuccess) return e;
    if ((e = c->shade.reserve(n)) != hipSuccess) return e;
    if ((e = c->walk_list.reserve(n)) != hipSuccess) return e;
    if (c->cfg.shading != SXE_SHADE_TEXTURE && (e = c->color.reserve(n)) != hipSuccess) return e;
    if ((e = c->fan_order.reserve(std::max<size_t>(c->fan_capacity, 1))) != hipSuccess) return e;
    return c->clip_list.reserve(c->fan_capacity);
}

// Region capacities are rounded up to 64 entries, so that a region starts on a 256-byte boundary.
void layout_from_caps(BinLayout &bl, const uint32_t cap[kBinRegions])
{
    uint32_t at = 0;
    for (int r = 0; r < kBinRegions; ++r) {
        bl.cap[r] = (cap[r] + 63u) & ~63u;
        bl.off[r] = at;
        at += bl.cap[r];
    }
    bl.stride = at;
}

// Makes every region at least `want` entries large (never smaller than it is) and sizes the bin array.
int set_bin_layout(sxe_ctx *c, const uint32_t want[kBinRegions])
{
    uint32_t cap[kBinRegions];
    uint64_t stride = 0;
    for (int r = 0; r < kBinRegions; ++r) {
        cap[r] = std::max(want[r], c->bl.cap[r]);
        stride += ((uint64_t)cap[r] + 63u) & ~(uint64_t)63u;
    }
    const uint64_t tiles = (uint64_t)std::max<uint32_t>(c->g.n_tiles, 1) * c->view_capacity;
    const uint64_t total = stride * tiles;
    if (total > 0xFFFFFFF0ull)
        return fail(c, SXE_E_CAPACITY, "bins of %llu entries per tile for %llu tiles exceed 2^32 entries", (unsigned long long)stride,
                    (unsigned long long)tiles);
    layout_from_caps(c->bl, cap);
    SXE_HIP(c, c->bins.reserve((size_t)total));
    return SXE_OK;
}

// Shared tail of update_drawables / update_instances: pools the meshes, derives the draw list
// (src/renderer/drawer.rs:173-214, 481-487, 313-347) and uploads everything.
int upload_scene(sxe_ctx *c, const std::vector<HostMesh> &hm, const std::vector<HostObject> &ho)
{
    SXE_HIP(c, hipSetDevice(c->cfg.device));
    // finish (and settle) whatever still reads the old scene
    {
        const int rc = sxe_sync(c);
        if (rc) return rc;
    }

    const bool need_tex = c->cfg.shading != SXE_SHADE_VERTEX_COLOR;
    std::vector<MeshDev> meshes(hm.size());
    size_t nv_total = 0, ni_total = 0;
    for (size_t m = 0; m < hm.size(); ++m) {
        if (hm[m].ni % 3) return fail(c, SXE_E_INVALID, "mesh %zu: index count %u is not a multiple of 3", m, hm[m].ni);
        if ((hm[m].nv && !hm[m].v) || (hm[m].ni && !hm[m].idx)) return fail(c, SXE_E_INVALID, "mesh %zu: null data", m);
        for (uint32_t k = 0; k < hm[m].ni; ++k)
            if (hm[m].idx[k] >= hm[m].nv) return fail(c, SXE_E_INVALID, "mesh %zu: index %u out of range", m, hm[m].idx[k]);
        meshes[m].vtx_base = (uint32_t)nv_total; // kept a multiple of 4 for 128-bit SoA loads
        meshes[m].n_vtx = hm[m].nv;
        meshes[m].idx_base = (uint32_t)ni_total;
        meshes[m].n_tri = hm[m].ni / 3;
        for (int k = 0; k < 3; ++k) {
            float lo = 0.0f, hi = 0.0f;
            for (uint32_t v = 0; v < hm[m].nv; ++v) {
                const float p = hm[m].v[v].pos[k];
                if (v == 0 || p < lo) lo = p;
                if (v == 0 || p > hi) hi = p;
                if (p != p) { lo = -INFINITY; hi = INFINITY; break; } // NaN coordinate: never cull this mesh as a whole
            }
            meshes[m].lo[k] = lo;
            meshes[m].hi[k] = hi;
        }
        nv_total += (hm[m].nv + 3u) & ~3u;
        ni_total += hm[m].ni;
        if (nv_total > 0xFFFFFFF0ull || ni_total > 0xFFFFFFF0ull) return fail(c, SXE_E_CAPACITY, "scene pools exceed 2^32 elements");
    }
    std::vector<sxe_vertex> verts(nv_total);
    std::vector<float> px(nv_total), py(nv_total), pz(nv_total);
    std::vector<uint32_t> idx(ni_total);
    if (nv_total) memset(verts.data(), 0, nv_total * sizeof(sxe_vertex));
    for (size_t m = 0; m < hm.size(); ++m) {
        if (hm[m].nv) memcpy(&verts[meshes[m].vtx_base], hm[m].v, hm[m].nv * sizeof(sxe_vertex));
        if (hm[m].ni) memcpy(&idx[meshes[m].idx_base], hm[m].idx, hm[m].ni * sizeof(uint32_t));
    }
    for (size_t v = 0; v < nv_total; ++v) {
        px[v] = verts[v].pos[0];
        py[v] = verts[v].pos[1];
        pz[v] = verts[v].pos[2];
    }

    // draw order: texture groups ascending, caller order inside a group; unrendered meshes skipped
    const bool compat = (c->cfg.flags & SXE_FLAG_COMPAT_REFERENCE_INDEXING) != 0;
    std::vector<uint32_t> order;
    order.reserve(ho.size());
    for (uint32_t o = 0; o < ho.size(); ++o) {
        if (ho[o].mesh >= hm.size()) return fail(c, SXE_E_INVALID, "object %u: mesh %u out of range", o, ho[o].mesh);
        // (the reference files an unrendered mesh under its texture like any other: it takes a
        // place in its group, src/renderer/drawer.rs:481-487,326-329)
        if (!ho[o].rendered && !compat) continue;
        if (need_tex && (ho[o].texture < 0 || (uint32_t)ho[o].texture >= c->n_textures))
            return fail(c, SXE_E_INVALID, "object %u: texture %d is not in the texture table (%u entries)", o,
                        ho[o].texture, c->n_textures);
        order.push_back(o);
    }
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return ho[a].texture < ho[b].texture; });
    int32_t max_texture = -1;
    if (need_tex)
        for (uint32_t o : order) max_texture = std::max(max_texture, ho[o].texture);

    // The draw list: which triangles (a mesh of the table), under which object's model matrix, with
    // which texture.  As the reference means it, every rendered object draws its own mesh under its
    // own matrix.
    struct HostDraw {
        uint32_t mesh, object;
        int32_t texture;
    };
    std::vector<HostDraw> draws;
    size_t box_meshes_from = SIZE_MAX; // entries of the mesh table from here on are boxes only: no triangles of their own
    std::vector<uint32_t> object_mesh(ho.size());
    for (size_t o = 0; o < ho.size(); ++o) object_mesh[o] = ho[o].mesh;
    if (!compat) {
        draws.reserve(order.size());
        for (uint32_t o : order) draws.push_back({ho[o].mesh, o, ho[o].texture});
    } else {
        // As the reference's command buffer records it (src/renderer/drawer.rs:313-347), flaws
        // included.  The vertex and index buffers hold the drawables' meshes one after the other in
        // caller order, indices rebased (:173-193).  Inside a texture group the i-th mesh, counting
        // the unrendered ones, is drawn with the uniform at offset i (:326,331): the i-th matrix
        // of the whole scene, not the mesh's own.  Its index range starts where the ranges of the
        // draws before it, in group order, end (:313,345-346), whichever drawable's indices lie
        // there, and an unrendered mesh moves that cursor by nothing (:327-329).
        std::vector<uint64_t> cat_off(ho.size() + 1, 0); // where each drawable's indices start in that index buffer
        for (size_t o = 0; o < ho.size(); ++o) cat_off[o + 1] = cat_off[o] + hm[ho[o].mesh].ni;
        // bounding box, per matrix, of everything drawn under it (what the whole-object visibility test needs)
        std::vector<MeshDev> boxes(ho.size());
        for (MeshDev &b : boxes) {
            b = MeshDev{};
            for (int k = 0; k < 3; ++k) { b.lo[k] = INFINITY; b.hi[k] = -INFINITY; }
        }
        uint64_t cursor = 0;
        uint32_t i = 0;
        for (size_t d = 0; d < order.size(); ++d, ++i) {
            const HostObject &o = ho[order[d]];
            if (d > 0 && ho[order[d - 1]].texture != o.texture) i = 0; // next group
            if (!o.rendered) continue;
            const uint32_t n_idx = hm[o.mesh].ni;
            // the drawable whose indices lie at the cursor (ranges never straddle the end: the
            // rendered meshes' index counts sum to at most the buffer's length)
            size_t k = (size_t)(std::upper_bound(cat_off.begin(), cat_off.end(), cursor) - cat_off.begin()) - 1;
            while (k < ho.size() && cat_off[k + 1] == cat_off[k]) ++k; // (drawables without indices own no position)
            uint32_t mesh;
            if (n_idx == 0 || (k < ho.size() && cat_off[k] == cursor && hm[ho[k].mesh].ni == n_idx)) {
                mesh = n_idx == 0 ? o.mesh : ho[k].mesh; // exactly one drawable's mesh (always, when all meshes are alike)
            } else {
                // a range that starts inside a mesh or runs over several: a mesh of its own over
                // the pooled vertices (index counts are multiples of 3, so triangles stay whole)
                MeshDev sm{};
                std::vector<uint32_t> at(n_idx); // pooled position of every index's vertex
                uint32_t vlo = 0xFFFFFFFFu, vhi = 0;
                size_t kk = k;
                for (uint32_t j = 0; j < n_idx; ++j) {
                    const uint64_t p = cursor + j;
                    while (p >= cat_off[kk + 1]) ++kk;
                    const uint32_t src = ho[kk].mesh;
                    at[j] = meshes[src].vtx_base + hm[src].idx[p - cat_off[kk]];
                    vlo = std::min(vlo, at[j]);
                    vhi = std::max(vhi, at[j]);
                }
                sm.vtx_base = vlo;
                sm.n_vtx = vhi - vlo + 1u;
                sm.idx_base = (uint32_t)idx.size();
                sm.n_tri = n_idx / 3;
                if ((uint64_t)idx.size() + n_idx > 0xFFFFFFF0ull) return fail(c, SXE_E_CAPACITY, "scene pools exceed 2^32 elements");
                for (int q = 0; q < 3; ++q) { sm.lo[q] = INFINITY; sm.hi[q] = -INFINITY; }
                for (uint32_t j = 0; j < n_idx; ++j) {
                    idx.push_back(at[j] - vlo);
                    for (int q = 0; q < 3; ++q) {
                        const float p = verts[at[j]].pos[q];
                        if (p != p) { sm.lo[q] = -INFINITY; sm.hi[q] = INFINITY; }
                        sm.lo[q] = std::min(sm.lo[q], p);
                        sm.hi[q] = std::max(sm.hi[q], p);
                    }
                }
                mesh = (uint32_t)meshes.size();
                meshes.push_back(sm);
            }
            draws.push_back({mesh, i, o.texture});
            MeshDev &b = boxes[i];
            b.n_tri += meshes[mesh].n_tri;
            for (int q = 0; q < 3; ++q) {
                b.lo[q] = std::min(b.lo[q], meshes[mesh].lo[q]);
                b.hi[q] = std::max(b.hi[q], meshes[mesh].hi[q]);
            }
            cursor += n_idx;
        }
        // every matrix is tested against the box of what is drawn under it
        box_meshes_from = meshes.size();
        for (size_t o = 0; o < ho.size(); ++o) {
            if (boxes[o].n_tri == 0)
                for (int q = 0; q < 3; ++q) boxes[o].lo[q] = boxes[o].hi[q] = 0.0f;
            object_mesh[o] = (uint32_t)meshes.size();
            meshes.push_back(boxes[o]);
        }
    }

    // vertex range touched by every run of 256 triangles of every mesh (mesh-local indices)
    std::vector<std::vector<std::pair<uint32_t, uint32_t>>> ranges(meshes.size());
    for (size_t m = 0; m < meshes.size(); ++m) {
        if (m >= box_meshes_from) break;
        const uint32_t nt = meshes[m].n_tri;
        const uint32_t *mi = idx.data() + meshes[m].idx_base;
        for (uint32_t t0 = 0; t0 < nt; t0 += kSetupThreads) {
            uint32_t lo = 0xFFFFFFFFu, hi = 0;
            const uint32_t t1 = std::min(nt, t0 + (uint32_t)kSetupThreads);
            for (uint32_t k = 3 * t0; k < 3 * t1; ++k) {
                lo = std::min(lo, mi[k]);
                hi = std::max(hi, mi[k]);
            }
            ranges[m].push_back({lo, hi - lo + 1u});
        }
    }

    std::vector<InstDev> insts(draws.size());
    std::vector<BlockDesc> blocks;
    uint64_t prims = 0;
    for (size_t d = 0; d < draws.size(); ++d) {
        const HostDraw &o = draws[d];
        insts[d].mesh = o.mesh;
        insts[d].object = o.object;
        insts[d].texture = o.texture;
        insts[d].prim_base = (uint32_t)prims;
        const uint32_t nt = meshes[o.mesh].n_tri;
        for (uint32_t t0 = 0; t0 < nt; t0 += kSetupThreads) {
            const auto &vr = ranges[o.mesh][t0 / kSetupThreads];
            blocks.push_back({(uint32_t)d, t0, vr.first, vr.second, o.object, meshes[o.mesh].vtx_base, meshes[o.mesh].idx_base,
                              nt, o.texture, (uint32_t)prims, 0u, 0u});
        }
        prims += nt;
        if (prims > SXE_MAX_PRIMS) return fail(c, SXE_E_CAPACITY, "more than %u triangles per frame", SXE_MAX_PRIMS);
        // record slots are 32-bit: 256 per work item plus the region for clipped fans
        if (blocks.size() > (1u << 23)) return fail(c, SXE_E_CAPACITY, "more than 2^23 setup work items (256 triangles each)");
    }
    // vertex-kernel work items and the clip-position layout: objects in caller order
    std::vector<VBlockDesc> vblocks;
    uint64_t inst_vtx = 0;
    for (uint32_t o = 0; o < ho.size(); ++o) {
        const uint32_t nv = meshes[ho[o].mesh].n_vtx;
        for (uint32_t v0 = 0; v0 < nv; v0 += 1024) vblocks.push_back({o, v0, ho[o].mesh, (uint32_t)inst_vtx});
        inst_vtx += nv;
        if (inst_vtx > 0xFFFFFFF0ull) return fail(c, SXE_E_CAPACITY, "more than 2^32 instance vertices");
    }

    std::vector<float> models(16 * ho.size());
    std::vector<sxe_transform> transforms(ho.size());
    for (size_t o = 0; o < ho.size(); ++o) {
        transforms[o] = ho[o].transform;
        sxe_transform_to_model(&ho[o].transform, &models[16 * o]);
    }

    auto up = [&](auto &buf, const auto &vec) -> hipError_t {
        hipError_t e = buf.reserve(std::max<size_t>(vec.size(), 1));
        if (e != hipSuccess || vec.empty()) return e;
        return hipMemcpy(buf.p, vec.data(), vec.size() * sizeof(vec[0]), hipMemcpyHostToDevice);
    };
    SXE_HIP(c, up(c->vertices, verts));
    SXE_HIP(c, up(c->pos_x, px));
    SXE_HIP(c, up(c->pos_y, py));
    SXE_HIP(c, up(c->pos_z, pz));
    SXE_HIP(c, up(c->indices, idx));
    SXE_HIP(c, up(c->meshes, meshes));
    SXE_HIP(c, up(c->insts, insts));
    SXE_HIP(c, up(c->blocks, blocks));
    SXE_HIP(c, up(c->vblocks, vblocks));
    SXE_HIP(c, up(c->models, models));
    SXE_HIP(c, up(c->transforms, transforms));
    SXE_HIP(c, up(c->object_mesh, object_mesh));
    if ((uint64_t)blocks.size() * kSetupThreads * c->view_capacity > 0x7FFFFFFFull) c->view_capacity = 1; // (slots are 32-bit)
    for (FrameSlot &fs : c->slots) {
        SXE_HIP(c, fs.visible.reserve(std::max<size_t>(ho.size(), 1) * c->view_capacity));
        SXE_HIP(c, fs.pvm.reserve(std::max<size_t>(models.size(), 16) * c->view_capacity));
        if (c->cfg.shading == SXE_SHADE_LIT) SXE_HIP(c, fs.models_snap.reserve(std::max<size_t>(models.size(), 16)));
    }
    SXE_HIP(c, c->blk_count.reserve(std::max<size_t>(blocks.size(), 1) * 4 * c->view_capacity)); // (a count per wave)

    c->n_meshes = (uint32_t)hm.size();
    c->n_objects = (uint32_t)ho.size();
    c->n_draws = (uint32_t)insts.size();
    c->n_blocks = (uint32_t)blocks.size();
    c->n_vblocks = (uint32_t)vblocks.size();
    c->n_prims = prims;
    c->n_inst_vertices = inst_vtx;
    c->have_transforms = true;
    c->have_scene = true;
    c->scene_max_texture = max_texture;
    c->scene_lost_textures = false;
    c->clip_enabled = true;
    c->clip_quiet = 0;

    // first guess for the work buffers; sxe_sync grows them and redraws when a frame needs more
    // every setup work item owns 256 record slots; fans of clipped triangles follow them
    if (c->fan_capacity < 4096) c->fan_capacity = 4096;
    SXE_HIP(c, reserve_records(c));
    // bins: room for four times the mean number of entries a tile would get if half of the
    // triangles survived, split 4 : 4 : 2 : 1 over the size classes
    {
        const uint64_t mean = prims / 2 / std::max<uint32_t>(c->g.n_tiles, 1);
        const uint32_t base = (uint32_t)std::min<uint64_t>(4096, std::max<uint64_t>(64, 4 * mean));
        const uint32_t want[kBinRegions] = {base, base, base / 2, base / 4, 64};
        const int rc = set_bin_layout(c, want);
        if (rc) return rc;
    }
    return SXE_OK;
}

// The work buffers of an ordinary frame: the context's own, with the slot's per-object matrices.
WorkBufs context_bufs(sxe_ctx *c, FrameSlot &fs)
{
    WorkBufs w;
    w.pvm = fs.pvm.p;
    w.visible = fs.visible.p;
    w.blk_count = c->blk_count.p;
    w.tile_counters = c->tile_counters.p;
    w.bins = c->bins.p;
    w.walk_list = c->walk_list.p;
    w.fan_order = c->fan_order.p;
    w.tile_launch = c->tile_launch.p;
    w.rast = c->rast.p;
    w.wide = c->wide.p;
    w.clip_list = c->clip_list.p;
    w.shade = c->shade.p;
    w.color = c->color.p;
    w.counters = c->counters.p;
    w.fan_capacity = c->fan_capacity;
    w.bl = c->bl;
    w.with_clip = c->clip_enabled;
    // a lit frame keeps the matrices it was lit with: a redraw after an overflow must not see what
    // the rotation system or update_uniforms have made of them since
    const bool lit = c->cfg.shading == SXE_SHADE_LIT;
    w.models_snap = lit ? fs.models_snap.p : nullptr;
    w.models_lit = lit ? fs.models_snap.p : c->models.p;
    return w;
}

ObjectArgs object_args(sxe_ctx *c, const WorkBufs &w)
{
    ObjectArgs oa{};
    oa.models = c->models.p;
    oa.pv = c->pv;
    oa.pvm = w.pvm;
    oa.meshes = c->meshes.p;
    oa.object_mesh = c->object_mesh.p;
    oa.visible = w.visible;
    oa.n = c->n_objects;
    oa.objects = 1;
    oa.zero = w.tile_counters;
    oa.n_zero = 0;
    oa.counters = w.counters;
    oa.g = c->g;
    oa.transforms = c->transforms.p;
    oa.models_rw = c->models.p;
    oa.models_snap = w.models_snap;
    oa.rot_seed = 0;
    oa.rot_first = oa.rot_n = 0;
    oa.rot_delta_ms = 0.0f;
    return oa;
}

// Copies `bytes` of caller memory to `dst` on the device, asynchronously on the context's stream.
int stage_upload(sxe_ctx *c, void *dst, const void *src, size_t bytes)
{
    if (bytes == 0) return SXE_OK;
    sxe_ctx::Staging &st = c->staging[c->staging_next++ % 4];
    if (!st.done) SXE_HIP(c, hipEventCreateWithFlags(&st.done, hipEventDisableTiming));
    if (st.used) SXE_HIP(c, hipEventSynchronize(st.done)); // its previous transfer has long finished
    if (st.cap < bytes) {
        if (st.p) SXE_HIP(c, hipHostFree(st.p));
        st.p = nullptr;
        st.cap = 0;
        SXE_HIP(c, hipHostMalloc(&st.p, bytes, hipHostMallocDefault));
        st.cap = bytes;
    }
    memcpy(st.p, src, bytes);
    // (on the context's own stream: a second stream for the transfers, ordered by events, was
    // measured slower -- the cross-stream waits cost more than the 25 us transfer they would hide)
    SXE_HIP(c, hipMemcpyAsync(dst, st.p, bytes, hipMemcpyHostToDevice, c->stream));
    SXE_HIP(c, hipEventRecord(st.done, c->stream));
    st.used = true;
    return SXE_OK;
}

// Launches a deferred rotation-system update on its own (the frame it was meant for is not the
// next thing to happen to the transforms).
int flush_rotation(sxe_ctx *c)
{
    if (!c->rot_pending) return SXE_OK;
    c->rot_pending = false;
    SXE_HIP(c, launch_rotation_system(c->transforms.p, c->models.p, c->n_objects, c->rot_seed, c->rot_first, c->rot_n,
                                      c->rot_delta_ms, c->stream));
    return SXE_OK;
}

// Per-kernel events cost a few microseconds each; a context may ask for them on a sample of the frames.
static bool frame_is_timed(const sxe_ctx *c, uint64_t frame)
{
    return (c->cfg.flags & SXE_FLAG_TIMING) != 0 && c->timing_stride != 0 && frame % c->timing_stride == 0;
}

// Enqueues the frame of slot `fs` with the work buffers `w`.  A redraw (fresh == false) keeps the
// per-object matrices in w.pvm.
int enqueue_frame(sxe_ctx *c, FrameSlot &fs, bool fresh, const WorkBufs &w)
{
    Image &img = c->images[fs.image];
    const bool timing = fs.timed;
    hipStream_t s = c->stream;
    FrameGeom g = c->g;
    g.views = fs.views;
    const uint32_t nt = total_tiles(g); // of all views of the set

    const size_t ntc = std::max<uint32_t>(nt, 1);
    if (timing) SXE_HIP(c, hipEventRecord(fs.ev[EV_BEGIN], s));
    c->last_tile_counters = w.tile_counters;
    if (fresh) fs.light = c->light;
    {
        ObjectArgs oa = object_args(c, w);
        oa.g = g;
        oa.pv_views = fs.views > 1 ? fs.pv_views.p : nullptr;
        oa.objects = fresh ? 1u : 0u;
        if (fresh && c->rot_pending) {
            oa.rot_seed = c->rot_seed;
            oa.rot_first = c->rot_first;
            oa.rot_n = c->rot_n;
            oa.rot_delta_ms = c->rot_delta_ms;
            c->rot_pending = false;
        }
        oa.n_zero = (uint32_t)((kTileCtrWords + 1) * ntc); // the TileCtr of every tile, then the late-fill counters
        SXE_HIP(c, launch_object(oa, s));
    }
    if (timing) SXE_HIP(c, hipEventRecord(fs.ev[EV_OBJECT], s));

    SetupArgs sa{};
    sa.vertices = c->vertices.p;
    sa.indices = c->indices.p;
    sa.meshes = c->meshes.p;
    sa.insts = c->insts.p;
    sa.blocks = c->blocks.p;
    sa.pvm = w.pvm;
    sa.models = w.models_lit;
    sa.light = fs.light;
    sa.visible = w.visible;
    sa.n_objects = c->n_objects;
    sa.n_blocks = c->n_blocks;
    sa.shading = c->cfg.shading;
    sa.debug = c->cfg.flags & kDebugMask;
    sa.g = g;
    sa.blk_count = w.blk_count;
    sa.tiles = reinterpret_cast<TileCtr *>(w.tile_counters);
    sa.bins = w.bins;
    sa.bl = w.bl;
    sa.textures = c->textab.p;
    sa.blend_scene = c->blend_scene ? 1u : 0u;
    sa.rast = w.rast;
    sa.wide = w.wide;
    sa.shade = w.shade;
    sa.color = w.color;
    sa.fan_capacity = (uint32_t)w.fan_capacity;
    sa.clip_list = w.clip_list;
    sa.walk_list = w.walk_list;
    sa.fan_order = w.fan_order;
    sa.counters = w.counters;
    sa.clip_off = w.with_clip ? 0u : 1u; // (k_setup must know whether k_clip follows)
    SXE_HIP(c, launch_setup(sa, s));
    if (w.with_clip) SXE_HIP(c, launch_clip(sa, s));
    if (timing) SXE_HIP(c, hipEventRecord(fs.ev[EV_SETUP], s));

    FinishArgs fa{};
    fa.g = g;
    fa.blk_count = w.blk_count;
    fa.n_blocks = c->n_blocks;
    fa.walk_blocks = 128; // 1024 waves share the list of late entries, however long it is
    fa.rast = w.rast;
    fa.wide = w.wide;
    fa.walk_list = w.walk_list;
    fa.fan_order = w.fan_order;
    fa.tiles = sa.tiles;
    fa.late_fill = w.tile_counters + kTileCtrWords * ntc;
    fa.tile_launch = w.tile_launch;
    fa.bins = w.bins;
    fa.bl = w.bl;
    fa.counters = w.counters;
    SXE_HIP(c, launch_finish(fa, s));
    if (timing) SXE_HIP(c, hipEventRecord(fs.ev[EV_FINISH], s));

    RasterArgs ra{};
    ra.g = g;
    ra.tile_launch = w.tile_launch;
    ra.bins = w.bins;
    ra.bl = w.bl;
    ra.rast = w.rast;
    ra.wide = w.wide;
    ra.shade = w.shade;
    ra.color = w.color;
    ra.fan_order = w.fan_order;
    ra.n_slots = fs.views * c->n_blocks * kSetupThreads;
    ra.image_stride = (size_t)c->cfg.width * c->cfg.height;
    ra.counters = w.counters;
    ra.footprints = c->footprints.p;
    ra.textures = c->textab.p;
    ra.n_textures = c->n_textures;
    ra.out_color = img.color;
    ra.out_depth = img.depth;
    ra.out_prim = (c->cfg.flags & SXE_FLAG_WRITE_COVERAGE) ? img.prim : nullptr;
    ra.clear_code = srgb_encode_host(c->cfg.clear[0]) | (srgb_encode_host(c->cfg.clear[1]) << 8) |
                    (srgb_encode_host(c->cfg.clear[2]) << 16) | (unorm8_host(c->cfg.clear[3]) << 24);
    ra.shading = c->cfg.shading;
    ra.tile_times = (c->cfg.flags & kDebugTileTimes) ? c->tile_times.p : nullptr;
    ra.debug = c->cfg.flags & kDebugMask;
    // the tile kernel writes the counters into the slot's pinned block itself, unless it does not
    // run or still changes them (development census)
    const bool direct = nt > 0 && fs.d_counters && !(c->cfg.flags & kDebugSkipAtomic);
    ra.host_counters = direct ? fs.d_counters : nullptr;
    ra.blend_scene = c->blend_scene ? 1u : 0u;
    ra.bgra = (c->cfg.flags & SXE_FLAG_BGRA) ? 1u : 0u;
    SXE_HIP(c, launch_raster_shade(ra, s));
    SXE_HIP(c, launch_raster_blend(ra, s)); // (only when a texture is translucent: the tiles the kernel above left out)
    if (timing) SXE_HIP(c, hipEventRecord(fs.ev[EV_RASTER], s));

    if (!direct) SXE_HIP(c, hipMemcpyAsync(fs.h_counters, w.counters, sizeof(FrameCounters), hipMemcpyDeviceToHost, s));
    SXE_HIP(c, hipEventRecord(fs.done, s));
    for (uint32_t v = 0; v < fs.views; ++v) c->images[fs.image + v].drawn = true;
    fs.pending = true;
    return SXE_OK;
}

// Grows the work buffers to what the frame in `fs` asked for.  Nothing may be in flight.
int grow_for(sxe_ctx *c, const FrameSlot &fs)
{
    const FrameCounters &k = *fs.h_counters;
    if (k.overflow & kOverflowNoClip) {
        c->clip_enabled = true;
        c->clip_quiet = 0;
    }
    // a full clip list hides how many fans its overflow would have produced: at most 7 each
    const size_t fans = std::max<size_t>(k.fan_records, k.clip_count > c->fan_capacity ? (size_t)k.clip_count * kMaxFan : 0);
    const size_t need_fan = fans + fans / 4 + 1024;
    if (need_fan > c->fan_capacity) {
        c->fan_capacity = need_fan;
        SXE_HIP(c, reserve_records(c));
    }
    uint32_t want[kBinRegions];
    bool grow = false;
    for (int r = 0; r < kBinRegions; ++r) {
        want[r] = k.need[r] > c->bl.cap[r] ? k.need[r] + k.need[r] / 4 + 16 : c->bl.cap[r];
        grow = grow || k.need[r] > c->bl.cap[r];
    }
    if (grow) return set_bin_layout(c, want);
    return SXE_OK;
}

void record_stats(sxe_ctx *c, const FrameCounters &k)
{
    c->stats.prims = c->n_prims;
    c->stats.records = k.total_records;
    c->stats.bin_entries = (uint32_t)std::min<unsigned long long>(k.total_pairs, 0xFFFFFFFFull);
    c->stats.tiles = c->g.n_tiles;
    c->stats.instance_vertices = c->n_inst_vertices;
    c->stats.objects = c->n_objects;
    c->stats.tile_size = kTile;
    c->stats.redraws = c->redraws;
}

// Cuts the work buffers of the context's scene out of this device's static array.  The bins get
// what the fixed-size parts leave.  False: the scene is too large for the array.
bool static_bufs(const sxe_ctx *c, unsigned char *base, WorkBufs &w, size_t &fixed_bytes)
{
    constexpr size_t kNoAllocFan = 4096; // fan records of clipped triangles, and clip list entries
    const size_t slots = (size_t)c->n_blocks * kSetupThreads, recs = slots + kNoAllocFan;
    const size_t ntc = std::max<uint32_t>(c->g.n_tiles, 1), nobj = std::max<uint32_t>(c->n_objects, 1);
    size_t at = 0;
    auto take = [&](size_t bytes) {
        unsigned char *p = base + at;
        at += (bytes + 255) & ~(size_t)255;
        return p;
    };
    w.counters = reinterpret_cast<FrameCounters *>(take(sizeof(FrameCounters)));
    w.blk_count = reinterpret_cast<uint32_t *>(take(16 * std::max<size_t>(c->n_blocks, 1)));
    w.tile_counters = reinterpret_cast<uint32_t *>(take(4 * (kTileCtrWords + 2) * ntc));
    w.tile_launch = reinterpret_cast<TileLaunch *>(take(sizeof(TileLaunch) * ntc));
    w.pvm = reinterpret_cast<float *>(take(64 * nobj));
    w.visible = reinterpret_cast<uint8_t *>(take(nobj));
    w.walk_list = reinterpret_cast<uint32_t *>(take(4 * recs));
    w.rast = reinterpret_cast<RastRec *>(take(sizeof(RastRec) * recs));
    w.wide = reinterpret_cast<WideRec *>(take(sizeof(WideRec) * recs));
    w.shade = reinterpret_cast<ShadeRec *>(take(sizeof(ShadeRec) * recs));
    w.color = c->cfg.shading != SXE_SHADE_TEXTURE ? reinterpret_cast<ColorRec *>(take(sizeof(ColorRec) * recs)) : nullptr;
    w.clip_list = reinterpret_cast<ClipItem *>(take(sizeof(ClipItem) * kNoAllocFan));
    w.fan_order = reinterpret_cast<uint32_t *>(take(4 * kNoAllocFan));
    fixed_bytes = at;
    // the bins get the rest: the same number of entries for every tile, split 4 : 4 : 2 : 1 : 1 over
    // the size classes and the late entries
    const size_t per_tile = at < kScratchBytes ? (kScratchBytes - at) / 4 / ntc : 0;
    if (per_tile < 12 * 64) return false;
    const uint32_t unit = (uint32_t)std::min<size_t>(per_tile / 12 / 64 * 64, 1u << 20);
    const uint32_t cap[kBinRegions] = {4 * unit, 4 * unit, 2 * unit, unit, unit};
    layout_from_caps(w.bl, cap);
    w.bins = reinterpret_cast<uint32_t *>(base + at);
    w.fan_capacity = kNoAllocFan;
    w.with_clip = true; // always: a frame that finds a triangle to clip cannot be drawn again here
    w.models_lit = c->models.p; // (the call returns when the frame is finished: nothing can change them meanwhile)
    w.models_snap = nullptr;
    return true;
}

// How many of `n` views go into one launch set: as many as keep the set's work buffers around
// 1 GiB (records of 100 bytes per slot, bins), at most 64, and within 32-bit slot numbers.
uint32_t views_per_set(const sxe_ctx *c, uint32_t n)
{
    const uint64_t slots = std::max<uint64_t>((uint64_t)c->n_blocks * kSetupThreads, 1);
    const uint64_t per_view = slots * 100 + (uint64_t)std::max<uint32_t>(c->g.n_tiles, 1) * (4ull * std::max<uint32_t>(c->bl.stride, 64) + 72) +
                              (uint64_t)c->n_objects * 65;
    uint64_t k = std::max<uint64_t>((1ull << 30) / per_view, 1);
    k = std::min<uint64_t>(k, 0x7FFFFFFFull / slots);
    k = std::min<uint64_t>(std::min<uint64_t>(k, 64), n);
    return (uint32_t)std::max<uint64_t>(k, 1);
}

// Sizes every per-frame buffer for launch sets of `views` views.  Nothing may be lost: frames in
// flight are taken out first.
int ensure_views(sxe_ctx *c, uint32_t views)
{
    if (views <= c->view_capacity) return SXE_OK;
    int rc = sxe_sync(c);
    if (rc) return rc;
    c->view_capacity = views;
    const size_t nt = (size_t)std::max<uint32_t>(c->g.n_tiles, 1) * views;
    for (FrameSlot &fs : c->slots) {
        SXE_HIP(c, fs.visible.reserve(std::max<size_t>(c->n_objects, 1) * views));
        SXE_HIP(c, fs.pvm.reserve(std::max<size_t>((size_t)c->n_objects * 16, 16) * views));
        SXE_HIP(c, fs.pv_views.reserve(16 * (size_t)views));
    }
    SXE_HIP(c, c->blk_count.reserve(std::max<size_t>(c->n_blocks, 1) * 4 * views));
    c->last_tile_counters = nullptr; // (the array may move)
    SXE_HIP(c, c->tile_counters.reserve((kTileCtrWords + 1) * nt));
    SXE_HIP(c, c->tile_launch.reserve(nt));
    SXE_HIP(c, reserve_records(c));
    return set_bin_layout(c, c->bl.cap);
}

// Takes the result of the oldest frame in flight: waits for it, and if one of its work buffers
// was too small, grows the buffers and renders it again -- and after it every younger frame in
// flight, in order, because they may share its image or have overflowed as well.
int retire_oldest(sxe_ctx *c)
{
    FrameSlot &fs = c->slots[c->frames_retired % kFrameSlots];
    SXE_HIP(c, hipEventSynchronize(fs.done));
    if (fs.h_counters->overflow) {
        SXE_HIP(c, hipStreamSynchronize(c->stream));
        bool hopeless = false; // a frame in flight that no regrowth can hold
        for (uint64_t f = c->frames_retired; f < c->frames_enqueued; ++f)
            hopeless = hopeless || c->slots[f % kFrameSlots].h_counters->total_pairs > 0xFFFFFFFFull; // (then a region has overflowed as well)
        if (hopeless) {
            for (FrameSlot &x : c->slots) x.pending = false;
            c->frames_retired = c->frames_enqueued;
            return fail(c, SXE_E_CAPACITY, "frame has more than 2^32 - 1 (triangle, tile) bin entries");
        }
        for (uint64_t f = c->frames_retired; f < c->frames_enqueued; ++f) {
            FrameSlot &r = c->slots[f % kFrameSlots];
            // the oldest frame overflowed; the younger ones are drawn again after it in any case
            for (int attempt = 0; attempt < 4 && (r.h_counters->overflow || (attempt == 0 && f > c->frames_retired)); ++attempt) {
                int rc = grow_for(c, r);
                if (rc) { // (out of memory, or bins beyond 2^32 entries) nothing in flight can be finished
                    for (FrameSlot &x : c->slots) x.pending = false;
                    c->frames_retired = c->frames_enqueued;
                    return rc;
                }
                c->redraws++;
                rc = enqueue_frame(c, r, false, context_bufs(c, r));
                if (rc) return rc;
                SXE_HIP(c, hipStreamSynchronize(c->stream));
            }
            if (r.h_counters->overflow) {
                // give up on everything in flight rather than report a half-drawn frame as good
                for (FrameSlot &x : c->slots) x.pending = false;
                c->frames_retired = c->frames_enqueued;
                return fail(c, SXE_E_CAPACITY, "frame still overflows its work buffers after regrowth");
            }
        }
    }
    fs.pending = false;
    c->frames_retired++;
    const FrameCounters &k = *fs.h_counters;
    constexpr uint32_t kClipQuiet = 8;
    if (k.clip_count) {
        c->clip_quiet = 0;
    } else if (c->clip_enabled && ++c->clip_quiet >= kClipQuiet) {
        c->clip_enabled = false;
    }
    record_stats(c, k);
    if ((c->cfg.flags & kDebugSkipAtomic) && getenv("SXE_DEBUG_COUNTERS"))
        fprintf(stderr, "[sxe] wide entries %u, wide box pixels %u, lane box pixels %u, clip %u fans %u\n", k.pad[0], k.pad[1],
                k.pad[2], k.clip_count, k.fan_records);
    if (fs.timed) {
        float ms[EV_N] = {};
        for (int i = 1; i < EV_N; ++i) SXE_HIP(c, hipEventElapsedTime(&ms[i], fs.ev[i - 1], fs.ev[i]));
        sxe_timings &t = c->timings;
        t.object_ms = ms[EV_OBJECT];
        t.setup_ms = ms[EV_SETUP];
        t.finish_ms = ms[EV_FINISH];
        t.raster_shade_ms = ms[EV_RASTER];
        SXE_HIP(c, hipEventElapsedTime(&t.frame_ms, fs.ev[EV_BEGIN], fs.ev[EV_RASTER]));
        t.transform_ms = 0.0f;
        if (fs.transform_timed) SXE_HIP(c, hipEventElapsedTime(&t.transform_ms, fs.ev_tr[0], fs.ev_tr[1]));
        fs.transform_timed = false;
        sxe_timings &s = c->timing_totals;
        s.transform_ms += t.transform_ms; s.object_ms += t.object_ms; s.setup_ms += t.setup_ms;
        s.finish_ms += t.finish_ms; s.raster_shade_ms += t.raster_shade_ms; s.frame_ms += t.frame_ms;
        c->timed_frames++;
    }
    return SXE_OK;
}

int check_ctx(sxe_ctx *c)
{
    if (!c) return fail(nullptr, SXE_E_INVALID, "null context");
    hipError_t e = hipSetDevice(c->cfg.device);
    if (e != hipSuccess) return fail(c, SXE_E_DEVICE, "hipSetDevice(%d) failed: %s", c->cfg.device, hipGetErrorString(e));
    return SXE_OK;
}

int read_image_plane(sxe_ctx *c, uint32_t image_index, void *dst, size_t size_bytes, int plane)
{
    int rc = check_ctx(c);
    if (rc) return rc;
    if (image_index >= c->images.size()) return fail(c, SXE_E_INVALID, "image index %u out of range", image_index);
    Image &img = c->images[image_index];
    if (!img.drawn) return fail(c, SXE_E_STATE, "image %u was never drawn", image_index);
    const size_t px = (size_t)c->cfg.width * c->cfg.height;
    const size_t need = px * 4;
    if (!dst || size_bytes < need) return fail(c, SXE_E_INVALID, "destination holds %zu bytes, %zu needed", size_bytes, need);
    rc = sxe_sync(c);
    if (rc) return rc;
    const void *src = plane == 0 ? (const void *)img.color : plane == 1 ? (const void *)img.depth : (const void *)img.prim;
    if (!src)
        return fail(c, SXE_E_STATE, plane == 1 ? "the depth images are not kept (SXE_FLAG_DISCARD_DEPTH)"
                                               : "coverage was not requested (SXE_FLAG_WRITE_COVERAGE)");
    // rows outside the scissor were never written: report them as cleared
    SXE_HIP(c, hipMemcpy(dst, src, need, hipMemcpyDeviceToHost));
    return SXE_OK;
}

} // namespace

extern "C" {

const char *sxe_version(void) { return "sxe-hip 0.1 (gfx950)"; }

void sxe_default_config(sxe_config *cfg, uint32_t width, uint32_t height)
{
    if (!cfg) return;
    memset(cfg, 0, sizeof *cfg);
    cfg->width = width;
    cfg->height = height;
    cfg->fov_deg = 45.0f;  // src/renderer/drawer.rs:249
    cfg->z_near = 0.1f;    // :251
    cfg->z_far = 1000.0f;  // :252
    cfg->clear[0] = 0.7f;  // :307
    cfg->clear[1] = 0.2f;
    cfg->clear[2] = 0.0f;
    cfg->clear[3] = 1.0f;
    cfg->device = 0;
    cfg->n_images = 2;
    cfg->scissor_y0 = 0;
    cfg->scissor_y1 = 0;
    cfg->shading = SXE_SHADE_TEXTURE;
    cfg->flags = 0;
}

const char *sxe_last_error(const sxe_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

static bool light_for_device(const sxe_light *light, sxe::LightDev &d);

int sxe_create(const sxe_config *cfg, sxe_ctx **out)
{
    if (!cfg || !out) return fail(nullptr, SXE_E_INVALID, "null argument");
    *out = nullptr;
    if (cfg->width == 0 || cfg->height == 0 || cfg->width > SXE_MAX_DIM || cfg->height > SXE_MAX_DIM)
        return fail(nullptr, SXE_E_INVALID, "frame size %ux%u outside 1..%u", cfg->width, cfg->height, SXE_MAX_DIM);
    if (cfg->shading > SXE_SHADE_LIT) return fail(nullptr, SXE_E_INVALID, "unknown shading mode %u", cfg->shading);
    if (!(cfg->fov_deg > 0.0f && cfg->fov_deg < 180.0f) || !(cfg->z_near > 0.0f) || !(cfg->z_far > cfg->z_near))
        return fail(nullptr, SXE_E_INVALID, "bad perspective parameters");
    sxe_config k = *cfg;
    if (k.scissor_y0 == 0 && k.scissor_y1 == 0) k.scissor_y1 = k.height;
    if (k.scissor_y0 > k.scissor_y1 || k.scissor_y1 > k.height)
        return fail(nullptr, SXE_E_INVALID, "scissor rows [%u,%u) outside the frame", k.scissor_y0, k.scissor_y1);
    // bits 26-31 select development builds of the kernels whose images are wrong on purpose
    if ((k.flags & kDebugMask) && !(getenv("SXE_DEVELOPMENT") && getenv("SXE_DEVELOPMENT")[0] == '1'))
        return fail(nullptr, SXE_E_INVALID, "flags 0x%x: bits 26-31 are development switches (set SXE_DEVELOPMENT=1 to use them)",
                    k.flags & kDebugMask);
    constexpr uint32_t kKnownFlags = SXE_FLAG_WRITE_COVERAGE | SXE_FLAG_TIMING | SXE_FLAG_COMPAT_REFERENCE_INDEXING | SXE_FLAG_BGRA | SXE_FLAG_DISCARD_DEPTH | kDebugMask;
    if (k.flags & ~kKnownFlags) return fail(nullptr, SXE_E_INVALID, "unknown flag bits 0x%x", k.flags & ~kKnownFlags);
    if (k.n_images == 0) k.n_images = 1;
    if (k.n_images > 65536) return fail(nullptr, SXE_E_CAPACITY, "at most 65536 images");

    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, SXE_E_DEVICE, "no HIP device available (%s); this library has no CPU path",
                    e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    if (k.device < 0 || k.device >= ndev) return fail(nullptr, SXE_E_INVALID, "device %d not in 0..%d", k.device, ndev - 1);
    if (k.device >= kMaxDevices) return fail(nullptr, SXE_E_CAPACITY, "device ordinals above %d are not supported", kMaxDevices - 1);
    e = hipSetDevice(k.device);
    if (e != hipSuccess) return fail(nullptr, SXE_E_DEVICE, "hipSetDevice failed: %s", hipGetErrorString(e));
    // per-device state of the library: the tile kernel's constant tables and the address of the
    // static work array of sxe_draw_noalloc
    if ((e = init_device_tables(k.device)) != hipSuccess)
        return fail(nullptr, SXE_E_DEVICE, "uploading the colour tables failed: %s", hipGetErrorString(e));
    if ((e = init_scratch(k.device)) != hipSuccess)
        return fail(nullptr, SXE_E_DEVICE, "hipGetSymbolAddress(scratch) failed: %s", hipGetErrorString(e));

    sxe_ctx *c = new (std::nothrow) sxe_ctx();
    if (!c) return fail(nullptr, SXE_E_DEVICE, "out of host memory");
    c->cfg = k;
    c->g = make_geom(k);
    c->view = identity();
    c->proj = identity();
    c->pv = identity();
    {
        sxe_light light;
        sxe_default_light(&light);
        light_for_device(&light, c->light);
    }
    auto bail = [&](const char *what, hipError_t err) {
        fail(nullptr, SXE_E_DEVICE, "%s failed: %s", what, hipGetErrorString(err));
        sxe_destroy(c);
        return SXE_E_DEVICE;
    };
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
    for (FrameSlot &fs : c->slots) {
        for (int i = 0; i < EV_N; ++i)
            if ((e = hipEventCreate(&fs.ev[i])) != hipSuccess) return bail("hipEventCreate", e);
        for (int i = 0; i < 2; ++i)
            if ((e = hipEventCreate(&fs.ev_tr[i])) != hipSuccess) return bail("hipEventCreate", e);
        if ((e = hipEventCreateWithFlags(&fs.done, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
        if ((e = hipHostMalloc(reinterpret_cast<void **>(&fs.h_counters), sizeof(FrameCounters), hipHostMallocDefault)) != hipSuccess)
            return bail("hipHostMalloc", e);
        memset(fs.h_counters, 0, sizeof(FrameCounters));
        if (hipHostGetDevicePointer(reinterpret_cast<void **>(&fs.d_counters), fs.h_counters, 0) != hipSuccess) {
            fs.d_counters = nullptr; // not mapped: the counters are copied instead
            (void)hipGetLastError();
        }
    }
    const size_t px = (size_t)k.width * k.height;
    c->images.resize(k.n_images);
    if ((e = c->color_pool.reserve(px * 4 * k.n_images)) != hipSuccess) return bail("hipMalloc(color)", e);
    const bool keep_depth = !(k.flags & SXE_FLAG_DISCARD_DEPTH);
    if (keep_depth && (e = c->depth_pool.reserve(px * k.n_images)) != hipSuccess) return bail("hipMalloc(depth)", e);
    if (k.flags & SXE_FLAG_WRITE_COVERAGE)
        if ((e = c->prim_pool.reserve(px * k.n_images)) != hipSuccess) return bail("hipMalloc(coverage)", e);
    // rows outside the scissor are never written by a frame: start them cleared
    if ((e = hipMemset(c->color_pool.p, 0, px * 4 * k.n_images)) != hipSuccess) return bail("hipMemset", e);
    if (keep_depth && (e = hipMemset(c->depth_pool.p, 0, px * 4 * k.n_images)) != hipSuccess) return bail("hipMemset", e);
    if (c->prim_pool.p && (e = hipMemset(c->prim_pool.p, 0xFF, px * 4 * k.n_images)) != hipSuccess) return bail("hipMemset", e);
    for (uint32_t i = 0; i < k.n_images; ++i) {
        c->images[i].color = c->color_pool.p + px * 4 * i;
        c->images[i].depth = keep_depth ? c->depth_pool.p + px * i : nullptr;
        c->images[i].prim = c->prim_pool.p ? c->prim_pool.p + px * i : nullptr;
    }
    const size_t nt = std::max<uint32_t>(c->g.n_tiles, 1);
    if ((e = c->tile_counters.reserve((kTileCtrWords + 1) * nt)) != hipSuccess) return bail("hipMalloc(tile counters)", e);
    if ((e = c->tile_launch.reserve(nt)) != hipSuccess) return bail("hipMalloc(tile launch records)", e);
    if ((e = c->counters.reserve(1)) != hipSuccess) return bail("hipMalloc(counters)", e);
    if ((k.flags & kDebugTileTimes) && (e = c->tile_times.reserve(8 * nt)) != hipSuccess) return bail("hipMalloc(tile times)", e);
    if ((e = c->texels.reserve(16)) != hipSuccess) return bail("hipMalloc(texels)", e);
    if ((e = c->footprints.reserve(2)) != hipSuccess) return bail("hipMalloc(footprints)", e);
    if ((e = c->textab.reserve(1)) != hipSuccess) return bail("hipMalloc(texture table)", e);
    *out = c;
    return SXE_OK;
}

void sxe_destroy(sxe_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->cfg.device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    c->vertices.release(); c->pos_x.release(); c->pos_y.release(); c->pos_z.release();
    c->indices.release(); c->meshes.release(); c->insts.release(); c->blocks.release(); c->vblocks.release();
    c->models.release(); c->transforms.release(); c->texels.release(); c->footprints.release(); c->textab.release();
    c->blk_count.release(); c->wide.release(); c->clip_list.release(); c->object_mesh.release(); c->tile_counters.release(); c->walk_list.release(); c->fan_order.release(); c->tile_launch.release();
    c->bins.release(); c->rast.release(); c->shade.release(); c->color.release(); c->counters.release();
    c->clip.release(); c->tile_times.release();
    c->color_pool.release(); c->depth_pool.release(); c->prim_pool.release();
    for (auto &img : c->images) {
        if (img.h_color) (void)hipHostFree(img.h_color);
    }
    for (auto &st : c->staging) {
        if (st.p) (void)hipHostFree(st.p);
        if (st.done) (void)hipEventDestroy(st.done);
    }
    for (FrameSlot &fs : c->slots) {
        fs.pvm.release(); fs.visible.release(); fs.pv_views.release(); fs.models_snap.release();
        if (fs.h_counters) (void)hipHostFree(fs.h_counters);
        for (int i = 0; i < EV_N; ++i) if (fs.ev[i]) (void)hipEventDestroy(fs.ev[i]);
        for (int i = 0; i < 2; ++i) if (fs.ev_tr[i]) (void)hipEventDestroy(fs.ev_tr[i]);
        if (fs.done) (void)hipEventDestroy(fs.done);
    }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int sxe_set_textures(sxe_ctx *c, const sxe_texture_desc *textures, uint32_t n)
{
    int rc = check_ctx(c);
    if (rc) return rc;
    if (n && !textures) return fail(c, SXE_E_INVALID, "null texture table");
    if (n > 65536) return fail(c, SXE_E_CAPACITY, "at most 65536 textures");
    std::vector<TexDev> tab(n);
    std::vector<size_t> texel_at(n);
    size_t total = 0, entries = 0;
    bool any_translucent = false;
    for (uint32_t t = 0; t < n; ++t) {
        const sxe_texture_desc &d = textures[t];
        if (!d.rgba || d.width == 0 || d.height == 0 || d.width > 16384 || d.height > 16384)
            return fail(c, SXE_E_INVALID, "texture %u: bad size %ux%u or null pixels", t, d.width, d.height);
        const size_t px = (size_t)d.width * d.height;
        // Opaque textures (every JPEG-derived one of the reference) are shaded once per pixel: with
        // alpha 1 the blend returns the source.  A texture with any other alpha sends the tiles its
        // triangles touch down the ordered path.
        bool translucent = false;
        for (size_t i = 0; i < px && !translucent; ++i) translucent = d.rgba[4 * i + 3] != 255;
        any_translucent = any_translucent || translucent;
        texel_at[t] = total;
        tab[t].offset = (uint32_t)entries;
        tab[t].width = d.width;
        tab[t].height = d.height;
        tab[t].flags = translucent ? kTexTranslucent : 0u;
        total += px;
        entries += ((size_t)d.width + 2) * ((size_t)d.height + 1);
        if (entries > 0xFFFFFFF0ull) return fail(c, SXE_E_CAPACITY, "texture pool exceeds 2^32 texels");
    }
    rc = sxe_sync(c); // frames in flight still sample the old table
    if (rc) return rc;
    SXE_HIP(c, c->texels.reserve(std::max<size_t>(total * 4, 16)));
    SXE_HIP(c, c->footprints.reserve(entries + 2));
    SXE_HIP(c, c->textab.reserve(std::max<size_t>(n, 1)));
    // The tile kernels sample tables of 8-byte cells (a texel and the one below it; two neighbouring
    // cells are a bilinear footprint), built here on the device from the texels as given.
    for (uint32_t t = 0; t < n; ++t) {
        uint8_t *texels = c->texels.p + 4 * texel_at[t];
        SXE_HIP(c, hipMemcpyAsync(texels, textures[t].rgba, 4 * (size_t)tab[t].width * tab[t].height, hipMemcpyHostToDevice, c->stream));
        SXE_HIP(c, launch_texture_footprints(texels, tab[t].width, tab[t].height, c->footprints.p + tab[t].offset, c->stream));
    }
    if (n) SXE_HIP(c, hipMemcpyAsync(c->textab.p, tab.data(), n * sizeof(TexDev), hipMemcpyHostToDevice, c->stream));
    SXE_HIP(c, hipStreamSynchronize(c->stream)); // (the caller's pixels and `tab` are only borrowed)
    c->n_textures = n;
    c->blend_scene = any_translucent;
    // The scene's texture indices were checked against the table of its upload.  A table that no
    // longer holds all of them must not be sampled with them: the scene has to be uploaded again.
    if (c->have_scene && c->scene_max_texture >= (int32_t)n) {
        c->have_scene = false;
        c->scene_lost_textures = true;
    }
    return SXE_OK;
}

int sxe_update_drawables(sxe_ctx *c, const sxe_drawable *drawables, uint32_t n)
{
    int rc = check_ctx(c);
    if (rc) return rc;
    rc = flush_rotation(c);
    if (rc) return rc;
    if (n && !drawables) return fail(c, SXE_E_INVALID, "null drawable array");
    std::vector<HostMesh> hm(n);
    std::vector<HostObject> ho(n);
    for (uint32_t i = 0; i < n; ++i) {
        hm[i] = {drawables[i].vertices, drawables[i].n_vertices, drawables[i].indices, drawables[i].n_indices};
        ho[i] = {i, drawables[i].texture, drawables[i].rendered, drawables[i].transform};
    }
    return upload_scene(c, hm, ho);
}

int sxe_update_instances(sxe_ctx *c, const sxe_mesh_desc *meshes, uint32_t n_meshes, const sxe_instance *instances,
                         uint32_t n_instances)
{
    int rc = check_ctx(c);
    if (rc) return rc;
    rc = flush_rotation(c);
    if (rc) return rc;
    if ((n_meshes && !meshes) || (n_instances && !instances)) return fail(c, SXE_E_INVALID, "null array");
    std::vector<HostMesh> hm(n_meshes);
    std::vector<HostObject> ho(n_instances);
    for (uint32_t i = 0; i < n_meshes; ++i) hm[i] = {meshes[i].vertices, meshes[i].n_vertices, meshes[i].indices, meshes[i].n_indices};
    for (uint32_t i = 0; i < n_instances; ++i)
        ho[i] = {instances[i].mesh, instances[i].texture, instances[i].rendered, instances[i].transform};
    return upload_scene(c, hm, ho);
}

int sxe_update_uniforms(sxe_ctx *c, const float *models, uint32_t n)
{
    int rc = check_ctx(c);
    if (rc) return rc;
    rc = flush_rotation(c);
    if (rc) return rc;
    if (!c->have_scene) return fail(c, SXE_E_STATE, "update_uniforms before update_drawables");
    if (n > c->n_objects) return fail(c, SXE_E_CAPACITY, "%u model matrices for %u objects", n, c->n_objects);
    if (n && !models) return fail(c, SXE_E_INVALID, "null matrices");
    rc = stage_upload(c, c->models.p, models, (size_t)n * 64);
    if (rc) return rc;
    c->have_transforms = false;
    return SXE_OK;
}

int sxe_update_transforms(sxe_ctx *c, const sxe_transform *transforms, uint32_t n)
{
    int rc = check_ctx(c);
    if (rc) return rc;
    rc = flush_rotation(c);
    if (rc) return rc;
    if (!c->have_scene) return fail(c, SXE_E_STATE, "update_transforms before update_drawables");
    if (n > c->n_objects) return fail(c, SXE_E_CAPACITY, "%u transforms for %u objects", n, c->n_objects);
    if (n && !transforms) return fail(c, SXE_E_INVALID, "null transforms");
    if (n) {
        rc = stage_upload(c, c->transforms.p, transforms, (size_t)n * sizeof(sxe_transform));
        if (rc) return rc;
        SXE_HIP(c, launch_transform_to_model(c->transforms.p, c->models.p, n, c->stream));
    }
    c->have_transforms = (n == c->n_objects) || c->have_transforms;
    return SXE_OK;
}

int sxe_rotation_system(sxe_ctx *c, uint64_t seed, uint32_t frame, float delta_ms)
{
    return sxe_rotation_system_span(c, seed, frame, 1, delta_ms);
}

int sxe_rotation_system_span(sxe_ctx *c, uint64_t seed, uint32_t first_frame, uint32_t n_frames, float delta_ms)
{
    const uint32_t frame = first_frame;
    int rc = check_ctx(c);
    if (rc) return rc;
    if (n_frames == 0) return SXE_OK;
    if (!c->have_scene) return fail(c, SXE_E_STATE, "rotation_system before update_drawables");
    if (!c->have_transforms) return fail(c, SXE_E_STATE, "model matrices were set directly; no transforms to rotate");
    // The update is applied by the first kernel of the frame drawn next; an update that is already
    // waiting goes first, on its own.
    rc = flush_rotation(c);
    if (rc) return rc;
    c->rot_pending = true;
    c->rot_seed = seed;
    c->rot_first = frame;
    c->rot_n = n_frames;
    c->rot_delta_ms = delta_ms;
    return SXE_OK;
}

int sxe_update_camera_matrices(sxe_ctx *c, const float *view, const float *proj)
{
    int rc = check_ctx(c);
    if (rc) return rc;
    if (!view || !proj) return fail(c, SXE_E_INVALID, "null matrix");
    memcpy(c->view.m, view, 64);
    memcpy(c->proj.m, proj, 64);
    c->pv = mul(c->proj, c->view); // standard.vert:28: proj * view (* model per object, on the device)
    c->have_camera = true;
    return SXE_OK;
}

int sxe_update_camera(sxe_ctx *c, const sxe_transform *camera)
{
    int rc = check_ctx(c);
    if (rc) return rc;
    if (!camera) return fail(c, SXE_E_INVALID, "null camera transform");
    float view[16], proj[16];
    // the reference derives the aspect from DIMS (src/renderer/drawer.rs:508); this build uses the target
    const float aspect = (float)c->cfg.width / (float)c->cfg.height;
    sxe_camera_matrices(camera, c->cfg.fov_deg, aspect, c->cfg.z_near, c->cfg.z_far, view, proj);
    return sxe_update_camera_matrices(c, view, proj);
}

int sxe_draw(sxe_ctx *c, uint32_t image_index)
{
    int rc = check_ctx(c);
    if (rc) return rc;
    if (!c->have_scene)
        return fail(c, SXE_E_STATE, c->scene_lost_textures ? "the texture table was replaced by one without all textures the scene "
                                                            "uses: call update_drawables again before draw"
                                                          : "draw before update_drawables");
    if (!c->have_camera) return fail(c, SXE_E_STATE, "draw before update_camera");
    if (image_index >= c->images.size()) return fail(c, SXE_E_INVALID, "image index %u out of range", image_index);
    // at most kFrameSlots frames are in flight: take the oldest one's result when all slots are used
    if (c->frames_enqueued - c->frames_retired >= (uint64_t)kFrameSlots) {
        rc = retire_oldest(c);
        if (rc) return rc;
    }
    FrameSlot &fs = c->slots[c->frames_enqueued % kFrameSlots];
    fs.image = image_index;
    fs.views = 1;
    fs.timed = frame_is_timed(c, c->frames_enqueued);
    rc = enqueue_frame(c, fs, true, context_bufs(c, fs));
    if (rc) return rc;
    c->frames_enqueued++;
    return SXE_OK;
}

int sxe_draw_views(sxe_ctx *c, const sxe_transform *cameras, uint32_t n, uint32_t first_image)
{
    int rc = check_ctx(c);
    if (rc) return rc;
    if (!c->have_scene) return fail(c, SXE_E_STATE, "draw before update_drawables");
    if (n && !cameras) return fail(c, SXE_E_INVALID, "null camera array");
    if ((uint64_t)first_image + n > c->images.size())
        return fail(c, SXE_E_INVALID, "views %u..%u do not fit the context's %zu images", first_image, first_image + n, c->images.size());
    if (n == 0) return SXE_OK;
    rc = flush_rotation(c); // the first kernel of a set of several views does not carry the update
    if (rc) return rc;
    const uint32_t per_set = views_per_set(c, n);
    rc = ensure_views(c, per_set);
    if (rc) return rc;
    const float aspect = (float)c->cfg.width / (float)c->cfg.height;
    std::vector<float> pv(16 * (size_t)per_set);
    for (uint32_t v0 = 0; v0 < n; v0 += per_set) {
        const uint32_t nv = std::min(per_set, n - v0);
        if (c->frames_enqueued - c->frames_retired >= (uint64_t)kFrameSlots) {
            rc = retire_oldest(c);
            if (rc) return rc;
        }
        FrameSlot &fs = c->slots[c->frames_enqueued % kFrameSlots];
        // update_camera (src/renderer/drawer.rs:507-518) for every view of the set
        for (uint32_t v = 0; v < nv; ++v) {
            float view[16], proj[16];
            sxe_camera_matrices(&cameras[v0 + v], c->cfg.fov_deg, aspect, c->cfg.z_near, c->cfg.z_far, view, proj);
            Mat4 V, P;
            memcpy(V.m, view, 64);
            memcpy(P.m, proj, 64);
            const Mat4 PV = mul(P, V);
            memcpy(&pv[16 * (size_t)v], PV.m, 64);
            if (v0 + v + 1 == n) { // the context's camera ends up as the last view's, as after n calls of update_camera
                c->view = V;
                c->proj = P;
                c->pv = PV;
                c->have_camera = true;
            }
        }
        SXE_HIP(c, fs.pv_views.reserve(16 * (size_t)c->view_capacity));
        rc = stage_upload(c, fs.pv_views.p, pv.data(), 64 * (size_t)nv);
        if (rc) return rc;
        fs.image = first_image + v0;
        fs.views = nv;
            fs.timed = frame_is_timed(c, c->frames_enqueued);
        if (nv == 1) c->pv = *reinterpret_cast<const Mat4 *>(&pv[0]); // (a set of one goes the single-view way)
        rc = enqueue_frame(c, fs, true, context_bufs(c, fs));
        if (rc) return rc;
        c->frames_enqueued++;
    }
    return SXE_OK;
}

int sxe_sync(sxe_ctx *c)
{
    int rc = check_ctx(c);
    if (rc) return rc;
    while (c->frames_retired < c->frames_enqueued) {
        rc = retire_oldest(c);
        if (rc) return rc;
    }
    SXE_HIP(c, hipStreamSynchronize(c->stream));
    return SXE_OK;
}

int sxe_draw_noalloc(sxe_ctx *c, uint32_t image_index)
{
    int rc = check_ctx(c);
    if (rc) return rc;
    if (!c->have_scene) return fail(c, SXE_E_STATE, "draw before update_drawables");
    if (!c->have_camera) return fail(c, SXE_E_STATE, "draw before update_camera");
    if (image_index >= c->images.size()) return fail(c, SXE_E_INVALID, "image index %u out of range", image_index);
    // taking a frame of sxe_draw out of flight may have to grow its buffers: not from here
    if (c->frames_retired != c->frames_enqueued)
        return fail(c, SXE_E_STATE, "frames of sxe_draw are in flight: call sxe_sync before sxe_draw_noalloc");
    unsigned char *base = nullptr;
    {
        std::lock_guard<std::mutex> hold(g_scratch_lock);
        base = g_scratch_ptr[c->cfg.device];
    }
    if (!base) return fail(c, SXE_E_DEVICE, "the static work array was not located on device %d", c->cfg.device);
    // one call at a time per GPU: two frames in the one array would overwrite each other's records
    const ScratchHold hold(c->cfg.device);
    if (!hold.mine)
        return fail(c, SXE_E_STATE, "another sxe_draw_noalloc call is using the static work array of device %d", c->cfg.device);
    WorkBufs w;
    size_t fixed = 0;
    if (!static_bufs(c, base, w, fixed)) {
        if (fixed < kScratchBytes)
            return fail(c, SXE_E_CAPACITY,
                        "the frame has %u screen tiles: after %zu bytes of records and tables the static work array of %zu "
                        "bytes leaves fewer than 768 bin entries for each of them (frames above about 10,000 tiles never fit)",
                        c->g.n_tiles, fixed, kScratchBytes);
        return fail(c, SXE_E_CAPACITY,
                    "scene needs %zu bytes of record and table space, the static work array holds %zu "
                    "(%u setup work items of 256 triangles, %u tiles, %u objects)",
                    fixed, kScratchBytes, c->n_blocks, c->g.n_tiles, c->n_objects);
    }
    // Nothing of an earlier call's contents is relied upon: the frame's first kernel clears the
    // counters and per-tile tables, and every other table entry is written before it is read.
    FrameSlot &fs = c->slots[c->frames_enqueued % kFrameSlots];
    fs.image = image_index;
    fs.views = 1;
    fs.timed = false;
    rc = enqueue_frame(c, fs, true, w);
    if (rc) return rc;
    SXE_HIP(c, hipEventSynchronize(fs.done));
    fs.pending = false; // (the slot was only borrowed: the frame counts of sxe_draw do not move)
    const FrameCounters &k = *fs.h_counters;
    if (k.overflow) {
        c->images[image_index].drawn = false;
        if (k.overflow & kOverflowBins)
            return fail(c, SXE_E_CAPACITY,
                        "frame needs bins of %u/%u/%u/%u/%u entries per tile (size classes 0-3, late entries), the static work "
                        "array has room for %u/%u/%u/%u/%u",
                        k.need[0], k.need[1], k.need[2], k.need[3], k.need[4], w.bl.cap[0], w.bl.cap[1], w.bl.cap[2], w.bl.cap[3],
                        w.bl.cap[4]);
        return fail(c, SXE_E_CAPACITY, "frame clips %u triangles into %u fan triangles, the static work array has room for %zu",
                    k.clip_count, k.fan_records, w.fan_capacity);
    }
    record_stats(c, k);
    return SXE_OK;
}

int sxe_get_timing_totals(sxe_ctx *c, sxe_timings *sums, uint64_t *frames)
{
    if (!c || !sums || !frames) return fail(c, SXE_E_INVALID, "null argument");
    if (!(c->cfg.flags & SXE_FLAG_TIMING)) return fail(c, SXE_E_STATE, "context was created without SXE_FLAG_TIMING");
    *sums = c->timing_totals;
    *frames = c->timed_frames;
    return SXE_OK;
}

int sxe_set_timing_stride(sxe_ctx *c, uint32_t every)
{
    if (!c) return fail(c, SXE_E_INVALID, "null context");
    c->timing_stride = every;
    return SXE_OK;
}

int sxe_reset_timing_totals(sxe_ctx *c)
{
    if (!c) return fail(c, SXE_E_INVALID, "null context");
    c->timing_totals = sxe_timings{};
    c->timed_frames = 0;
    return SXE_OK;
}

int sxe_read_color(sxe_ctx *c, uint32_t image_index, uint8_t *rgba, size_t size_bytes)
{
    return read_image_plane(c, image_index, rgba, size_bytes, 0);
}

int sxe_map_color(sxe_ctx *c, uint32_t image_index, const uint8_t **rgba)
{
    int rc = check_ctx(c);
    if (rc) return rc;
    if (!rgba) return fail(c, SXE_E_INVALID, "null result pointer");
    if (image_index >= c->images.size()) return fail(c, SXE_E_INVALID, "image index %u out of range", image_index);
    Image &img = c->images[image_index];
    if (!img.drawn) return fail(c, SXE_E_STATE, "image %u was never drawn", image_index);
    const size_t need = (size_t)c->cfg.width * c->cfg.height * 4;
    if (!img.h_color) SXE_HIP(c, hipHostMalloc(reinterpret_cast<void **>(&img.h_color), need, hipHostMallocDefault));
    rc = sxe_sync(c);
    if (rc) return rc;
    SXE_HIP(c, hipMemcpyAsync(img.h_color, img.color, need, hipMemcpyDeviceToHost, c->stream));
    SXE_HIP(c, hipStreamSynchronize(c->stream));
    *rgba = img.h_color;
    return SXE_OK;
}

int sxe_read_depth(sxe_ctx *c, uint32_t image_index, float *depth, size_t size_bytes)
{
    return read_image_plane(c, image_index, depth, size_bytes, 1);
}

int sxe_read_coverage(sxe_ctx *c, uint32_t image_index, uint32_t *prim_ids, size_t size_bytes)
{
    return read_image_plane(c, image_index, prim_ids, size_bytes, 2);
}

int sxe_read_clip_positions(sxe_ctx *c, float *clip, size_t size_bytes)
{
    int rc = check_ctx(c);
    if (rc) return rc;
    rc = flush_rotation(c);
    if (rc) return rc;
    if (!c->have_scene || !c->have_camera) return fail(c, SXE_E_STATE, "read_clip_positions needs a scene and a camera");
    const size_t need = (size_t)c->n_inst_vertices * 16;
    if (!clip || size_bytes < need) return fail(c, SXE_E_INVALID, "destination holds %zu bytes, %zu needed", size_bytes, need);
    rc = sxe_sync(c); // the matrices go through a frame slot: none may be in use
    if (rc) return rc;
    SXE_HIP(c, c->clip.reserve(std::max<size_t>(c->n_inst_vertices, 1)));
    FrameSlot &fs = c->slots[c->frames_enqueued % kFrameSlots];
    SXE_HIP(c, launch_object(object_args(c, context_bufs(c, fs)), c->stream));
    SXE_HIP(c, launch_vertex(c->pos_x.p, c->pos_y.p, c->pos_z.p, c->meshes.p, c->vblocks.p, c->n_vblocks, fs.pvm.p,
                             c->clip.p, c->stream));
    SXE_HIP(c, hipStreamSynchronize(c->stream));
    if (need) SXE_HIP(c, hipMemcpy(clip, c->clip.p, need, hipMemcpyDeviceToHost));
    return SXE_OK;
}

int sxe_read_models(sxe_ctx *c, float *models, size_t size_bytes)
{
    int rc = check_ctx(c);
    if (rc) return rc;
    rc = flush_rotation(c);
    if (rc) return rc;
    if (!c->have_scene) return fail(c, SXE_E_STATE, "no scene");
    const size_t need = (size_t)c->n_objects * 64;
    if (!models || size_bytes < need) return fail(c, SXE_E_INVALID, "destination holds %zu bytes, %zu needed", size_bytes, need);
    SXE_HIP(c, hipStreamSynchronize(c->stream));
    if (need) SXE_HIP(c, hipMemcpy(models, c->models.p, need, hipMemcpyDeviceToHost));
    return SXE_OK;
}

int sxe_read_transforms(sxe_ctx *c, sxe_transform *transforms, size_t size_bytes)
{
    int rc = check_ctx(c);
    if (rc) return rc;
    rc = flush_rotation(c);
    if (rc) return rc;
    if (!c->have_scene || !c->have_transforms) return fail(c, SXE_E_STATE, "no device-resident transforms");
    const size_t need = (size_t)c->n_objects * sizeof(sxe_transform);
    if (!transforms || size_bytes < need) return fail(c, SXE_E_INVALID, "destination holds %zu bytes, %zu needed", size_bytes, need);
    SXE_HIP(c, hipStreamSynchronize(c->stream));
    if (need) SXE_HIP(c, hipMemcpy(transforms, c->transforms.p, need, hipMemcpyDeviceToHost));
    return SXE_OK;
}

int sxe_color_device_ptr(sxe_ctx *c, uint32_t image_index, void **ptr)
{
    if (!c) return fail(nullptr, SXE_E_INVALID, "null context");
    if (!ptr || image_index >= c->images.size()) return fail(c, SXE_E_INVALID, "bad image index or null pointer");
    *ptr = c->images[image_index].color;
    return SXE_OK;
}

int sxe_stream(sxe_ctx *c, void **stream)
{
    if (!c) return fail(nullptr, SXE_E_INVALID, "null context");
    if (!stream) return fail(c, SXE_E_INVALID, "null pointer");
    *stream = reinterpret_cast<void *>(c->stream);
    return SXE_OK;
}

int sxe_read_tile_counts(sxe_ctx *c, uint32_t *counts, size_t size_bytes)
{
    int rc = check_ctx(c);
    if (rc) return rc;
    const size_t need = (size_t)c->g.n_tiles * 4;
    if (!counts || size_bytes < need) return fail(c, SXE_E_INVALID, "destination holds %zu bytes, %zu needed", size_bytes, need);
    rc = sxe_sync(c);
    if (rc) return rc;
    if (!c->last_tile_counters) return fail(c, SXE_E_STATE, "no frame has been drawn");
    if (need) {
        std::vector<TileCtr> ctr(c->g.n_tiles);
        SXE_HIP(c, hipMemcpy(ctr.data(), c->last_tile_counters, ctr.size() * sizeof(TileCtr), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < ctr.size(); ++i) counts[i] = bin_size(ctr[i]);
    }
    return SXE_OK;
}

int sxe_read_tile_parts(sxe_ctx *c, uint32_t *parts, size_t size_bytes)
{
    int rc = check_ctx(c);
    if (rc) return rc;
    const size_t need = (size_t)c->g.n_tiles * 20;
    if (!parts || size_bytes < need) return fail(c, SXE_E_INVALID, "destination holds %zu bytes, %zu needed", size_bytes, need);
    rc = sxe_sync(c);
    if (rc) return rc;
    if (!c->last_tile_counters) return fail(c, SXE_E_STATE, "no frame has been drawn");
    if (need) {
        std::vector<TileCtr> ctr(c->g.n_tiles);
        SXE_HIP(c, hipMemcpy(ctr.data(), c->last_tile_counters, ctr.size() * sizeof(TileCtr), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < ctr.size(); ++i) {
            for (int k = 0; k < kSizeClasses; ++k) parts[5 * i + k] = ctr[i].cnt[k];
            parts[5 * i + 4] = ctr[i].multi;
        }
    }
    return SXE_OK;
}

int sxe_read_tile_times(sxe_ctx *c, uint64_t *clocks, size_t size_bytes)
{
    int rc = check_ctx(c);
    if (rc) return rc;
    if (!c->tile_times.p) return fail(c, SXE_E_STATE, "tile clocks were not requested when the context was created");
    const size_t need = (size_t)c->g.n_tiles * 64;
    if (!clocks || size_bytes < need) return fail(c, SXE_E_INVALID, "destination holds %zu bytes, %zu needed", size_bytes, need);
    rc = sxe_sync(c);
    if (rc) return rc;
    if (need) SXE_HIP(c, hipMemcpy(clocks, c->tile_times.p, need, hipMemcpyDeviceToHost));
    return SXE_OK;
}

int sxe_get_timings(sxe_ctx *c, sxe_timings *out)
{
    if (!c) return fail(nullptr, SXE_E_INVALID, "null context");
    if (!out) return fail(c, SXE_E_INVALID, "null output");
    if (!(c->cfg.flags & SXE_FLAG_TIMING)) return fail(c, SXE_E_STATE, "context was created without SXE_FLAG_TIMING");
    *out = c->timings;
    return SXE_OK;
}

int sxe_get_stats(sxe_ctx *c, sxe_frame_stats *out)
{
    if (!c) return fail(nullptr, SXE_E_INVALID, "null context");
    if (!out) return fail(c, SXE_E_INVALID, "null output");
    *out = c->stats;
    return SXE_OK;
}

// ---- host-side Transform / camera maths ----

void sxe_transform_identity(sxe_transform *t)
{
    const Quat q = quat_from_euler_deg(0.0f, 0.0f, 0.0f);
    t->position[0] = t->position[1] = t->position[2] = 0.0f;
    t->scale[0] = t->scale[1] = t->scale[2] = 1.0f;
    t->rotation[0] = q.s; t->rotation[1] = q.x; t->rotation[2] = q.y; t->rotation[3] = q.z;
}

void sxe_transform_translate(sxe_transform *t, float x, float y, float z)
{
    t->position[0] = t->position[0] + x;
    t->position[1] = t->position[1] + y;
    t->position[2] = t->position[2] + z;
}

void sxe_transform_rotate(sxe_transform *t, float x, float y, float z)
{
    const Quat by = quat_from_euler_deg(x, y, z);
    const Quat q0 = {t->rotation[0], t->rotation[1], t->rotation[2], t->rotation[3]};
    const Quat q = mul(q0, by);
    t->rotation[0] = q.s; t->rotation[1] = q.x; t->rotation[2] = q.y; t->rotation[3] = q.z;
}

void sxe_transform_to_model(const sxe_transform *t, float model[16])
{
    const Quat q = {t->rotation[0], t->rotation[1], t->rotation[2], t->rotation[3]};
    const Mat4 M = model_matrix(t->position, t->scale, q);
    memcpy(model, M.m, 64);
}

void sxe_default_light(sxe_light *light)
{
    if (!light) return;
    light->to_light[0] = 0.3f; light->to_light[1] = 0.8f; light->to_light[2] = 0.5f;
    light->to_viewer[0] = 0.0f; light->to_viewer[1] = 0.0f; light->to_viewer[2] = 1.0f;
    light->ambient = 0.2f;
    light->diffuse = 0.7f;
    light->specular = 0.3f;
    light->shininess_log2 = 5u;
}

static bool unit_vector(const float v[3], float out[3])
{
    const float len = sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
    if (!(len > 0.0f) || !(len <= 3.0e38f)) return false;
    out[0] = v[0] / len; out[1] = v[1] / len; out[2] = v[2] / len;
    return true;
}

int sxe_light_vectors(const sxe_light *light, float to_light[3], float to_viewer[3], float half[3])
{
    if (!light || !to_light || !to_viewer || !half || light->shininess_log2 > 16u) return SXE_E_INVALID;
    if (!unit_vector(light->to_light, to_light) || !unit_vector(light->to_viewer, to_viewer)) return SXE_E_INVALID;
    const float sum[3] = {to_light[0] + to_viewer[0], to_light[1] + to_viewer[1], to_light[2] + to_viewer[2]};
    if (!unit_vector(sum, half)) half[0] = half[1] = half[2] = 0.0f; // (light straight behind the viewer's back: no highlight)
    return SXE_OK;
}

static bool light_for_device(const sxe_light *light, sxe::LightDev &d)
{
    float l[3], v[3], h[3];
    if (sxe_light_vectors(light, l, v, h) != SXE_OK) return false;
    if (!(fabsf(light->ambient) <= 3.0e38f) || !(fabsf(light->diffuse) <= 3.0e38f) || !(fabsf(light->specular) <= 3.0e38f)) return false;
    d.lx = l[0]; d.ly = l[1]; d.lz = l[2];
    d.hx = h[0]; d.hy = h[1]; d.hz = h[2];
    d.vx = v[0]; d.vy = v[1]; d.vz = v[2];
    d.ambient = light->ambient; d.diffuse = light->diffuse; d.specular = light->specular;
    d.shininess_log2 = light->shininess_log2;
    return true;
}

int sxe_set_light(sxe_ctx *c, const sxe_light *light)
{
    int rc = check_ctx(c);
    if (rc) return rc;
    sxe::LightDev d;
    if (!light || !light_for_device(light, d))
        return fail(c, SXE_E_INVALID, "light: directions must be finite and not zero, the terms finite, shininess_log2 at most 16");
    c->light = d; // (kernel arguments of the frames drawn from now on)
    return SXE_OK;
}

void sxe_camera_matrices(const sxe_transform *camera, float fov_deg, float aspect, float z_near, float z_far,
                         float view[16], float proj[16])
{
    const Quat q = {camera->rotation[0], camera->rotation[1], camera->rotation[2], camera->rotation[3]};
    float ex, ey, ez;
    euler_from_quat(q, ex, ey, ez);
    (void)ez; // roll is ignored by the reference
    // argument order of src/renderer/drawer.rs:246: pitch = euler.y, yaw = euler.x
    const Mat4 V = fps_view(camera->position, ey, ex);
    const Mat4 P = perspective_yflip(fov_deg, aspect, z_near, z_far);
    memcpy(view, V.m, 64);
    memcpy(proj, P.m, 64);
}

} // extern "C"
