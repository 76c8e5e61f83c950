// sxe_internal.h -- device-side data layout and kernel launch interface of libsxe_hip.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sxe.h"
#include "sxe_math.h"

namespace sxe {

constexpr int kTile = 32;          // screen tile edge, pixels: one workgroup rasterises one tile
constexpr int kTilePixels = kTile * kTile;
constexpr int kSetupThreads = 256; // triangles per setup workgroup = record slots per segment
constexpr int kRasterThreads = 256;
constexpr float kGuard = 4.0f;     // guard band: x,y are clipped at +-kGuard*w (power of two)
constexpr int kMaxFan = 7;         // a triangle clipped by six planes has at most 9 corners
constexpr int kLaneBox = 24;       // largest pixel box one lane walks alone; larger ones are shared by lanes,
constexpr int kLaneBoxCrowd = 64;  // unless at least kCrowd lanes of the wave hold boxes up to this size
constexpr int kCrowd = 24;
// Untouched pixel of the tile's (depth bits << 32 | order) buffer: depth 1.0, order 0.  A fragment
// at depth >= 1.0 can never be Less than the cleared 1.0, and its key is never below this value, so
// the 64-bit minimum rejects it without a comparison of its own.
constexpr unsigned long long kVisEmpty = 0x3F800000ull << 32;
constexpr int32_t kNoRaster = INT32_MIN; // RastRec.x0 of a slot that holds no triangle to rasterise
// development-only switches carried in sxe_config.flags (timing experiments; outputs are wrong)
constexpr uint32_t kDebugSkipRaster = 1u << 30, kDebugSkipResolve = 1u << 31;
constexpr uint32_t kDebugSkipSmall = 1u << 28, kDebugSkipLarge = 1u << 29, kDebugSkipAtomic = 1u << 27;
constexpr uint32_t kDebugTileTimes = 1u << 26; // record start/end clock of every tile workgroup
constexpr uint32_t kDebugMask = 63u << 26;

// overflow bits of FrameCounters::overflow
constexpr uint32_t kOverflowBins = 1u, kOverflowClip = 2u;
constexpr uint32_t kOverflowNoClip = 4u; // a triangle needs clipping in a frame enqueued without the clip kernel
constexpr uint32_t kOverflowPairs = 8u;  // more than 2^32 - 1 (triangle, tile) pairs: beyond any bin array (found by the host)
// Tiles that hold a triangle with a translucent texture are rendered by k_raster_blend: every
// fragment is depth-tested and blended in submission order (BlendState::ALPHA,
// src/renderer/drawer.rs:660-663), which takes the tile's entries sorted, this many at a time.
constexpr uint32_t kBlendMaxEntries = 4096;
constexpr uint32_t kTileBlend = 1u;      // TileCtr::flags

struct MeshDev {
    uint32_t vtx_base, n_vtx, idx_base, n_tri;
    float lo[3], hi[3]; // object-space bounding box of the mesh
};

// One drawn object in draw order (texture groups ascending, caller order inside a group;
// src/renderer/drawer.rs:315-347,481-487).
struct InstDev {
    uint32_t mesh;
    uint32_t object;   // caller-order index: selects the model matrix
    int32_t texture;
    uint32_t prim_base; // id of this object's first triangle in the frame
};

// Work item of the setup kernel: 256 consecutive triangles of one object.  It repeats what the
// kernel needs from the object's InstDev and MeshDev so that a workgroup starts with one load
// instead of a chain of three.
struct __attribute__((aligned(16))) BlockDesc {
    uint32_t inst, tri_start;
    uint32_t vtx_lo, vtx_n; // range of mesh-local vertex indices those triangles use
    uint32_t object, vtx_base, idx_base, n_tri;
    int32_t texture;
    uint32_t prim_base, pad0, pad1;
};
constexpr int kVtxCache = 512; // vertices a setup workgroup can project once and share through LDS

struct VBlockDesc { // work item of the vertex kernel: up to 1024 consecutive vertices of one object
    uint32_t object, vtx_start, mesh, clip_base;
};

struct TexDev {
    uint32_t offset; // first cell of the texture's table in the pooled tables
    uint32_t width, height;
    uint32_t flags;  // kTexTranslucent
};
constexpr uint32_t kTexTranslucent = 1u; // some texel has alpha below 255: fragments must be blended in submission order

// Record storage.  Setup workgroup b owns slots [256 b, 256 b + 256), 64 per wave, and every wave
// fills its own from the front in submission order, so a slot number orders triangles as they were submitted.  Triangles cut by
// the clipper keep one such slot (their "order") and put their fan triangles into the extra
// region that starts at slot 256 * n_blocks.

// Everything the coverage/depth loop needs about one screen triangle (32 B, two 16-B loads).
// Corners are 24.8 fixed point, oriented so that the doubled area is > 0.  Corners 1 and 2 are
// stored as 16-bit offsets from corner 0 (x in the low half, y in the high half) when the triangle
// is narrow (its box spans fewer than 16384 sub-pixel units each way: every edge function value at
// a pixel of the box then fits 32 bits and every product 24-bit factors); any other triangle sets
// d1 = kWideTri and keeps its corners in the WideRec of the same slot.
// The low half of a fragment's visibility key, the slot that ranks its triangle in submission
// order, is the record's own slot number, except for the fan triangles of a clipped primitive,
// which rank as their primitive does: fan_order[] holds that slot for them.
struct __attribute__((aligned(16))) RastRec {
    int32_t x0, y0;   // x0 == kNoRaster: nothing to rasterise in this slot
    uint32_t d1, d2;
    float z0, dz1, dz2;
    float rA;         // 1 / float(doubled area): one division per triangle instead of one per bin entry and pixel
};
constexpr uint32_t kWideTri = 0x80008000u; // both offsets -32768: never produced by a packed triangle

struct __attribute__((aligned(16))) WideRec {
    int32_t x1, y1, x2, y2;
};

// Perspective-correct attribute planes of the same triangle, read once per visible pixel.
struct __attribute__((aligned(16))) ShadeRec {
    float iw0, diw1, diw2, uw0;
    float duw1, duw2, vw0, dvw1;
    float dvw2;
    uint32_t prim;       // primitive id reported in the coverage buffer
    uint32_t fan_or_ra;  // the order slot of a clipped triangle (fan count != 0): its first fan record;
                         // every other record: the bits of RastRec::rA, so that shading needs no second load
    uint32_t tex_fan;    // texture index in the low 24 bits, fan triangle count in the high 8 (0: ordinary)
};

struct __attribute__((aligned(16))) ColorRec {
    float rw0, drw1, drw2, gw0;
    float dgw1, dgw2, bw0, dbw1;
    float dbw2;
    float lit_a, lit_s; // SXE_SHADE_LIT: the triangle's two lighting terms, rgb = texture.rgb * lit_a + lit_s
    float pad2;
};

// The light of SXE_SHADE_LIT as the setup kernels take it: unit vectors in world space.
struct LightDev {
    float lx, ly, lz; // towards the light
    float hx, hy, hz; // half vector between light and viewer
    float vx, vy, vz; // towards the viewer
    float ambient, diffuse, specular;
    uint32_t shininess_log2;
};

struct FrameCounters {
    uint32_t total_records; // surviving triangles (order slots)
    uint32_t unused0;
    uint32_t fan_records;   // fan triangles of clipped primitives
    uint32_t overflow;      // kOverflow* bits
    uint32_t clip_count;    // triangles handed to the clip kernel
    uint32_t walk_count;    // triangles that take their bin entries in k_finish (entries of walk_list)
    uint32_t need[5];       // per bin region: the largest number of entries any tile has for it
    uint32_t blend_max;     // most entries of any tile that takes the ordered (blending) path
    uint32_t pad[4];        // development counters (box census)
    uint32_t unused1;
    unsigned long long total_pairs; // (triangle, tile) bin entries of all tiles (may exceed 32 bits: the host checks)
};

// A triangle that crosses the near, far or a guard-band plane: k_setup reserves its order slot and
// leaves the clipping to k_clip.
struct ClipItem {
    uint32_t inst, tri, slot, view;
};

// One launch set renders `views` views of the scene, all of the same size, each with its own camera
// into its own image (sxe_draw: one).  View v owns tiles [v * n_tiles, (v + 1) * n_tiles) of the
// per-tile tables and bins, setup work items and record slots [v * n_blocks, (v + 1) * n_blocks)
// (times 256 slots), and objects [v * n_objects, (v + 1) * n_objects) of the per-object arrays.
struct FrameGeom {
    uint32_t width, height;
    uint32_t sy0, sy1;       // scissor rows [sy0, sy1)
    uint32_t tiles_x, tile_row0, n_tile_rows, n_tiles; // n_tiles: per view
    float half_w, half_h;
    uint32_t views;
};
__host__ __device__ inline uint32_t total_tiles(const FrameGeom &g) { return g.n_tiles * g.views; }

struct ObjectArgs {
    const float *models;
    Mat4 pv;                     // proj * view of a single view
    const float *pv_views;       // the same per view (16 floats each) when g.views > 1, else null
    float *pvm;
    const MeshDev *meshes;
    const uint32_t *object_mesh; // mesh of every object (caller order)
    uint8_t *visible;            // 0: the whole object is outside the view volume or the scissor band
    uint32_t n;
    uint32_t objects;            // 0: only clear the counters (a frame drawn again keeps its matrices)
    uint32_t *zero;              // per-tile counters to clear
    uint32_t n_zero;
    FrameCounters *counters;     // cleared as well
    FrameGeom g;
    // Rotation::run (src/systems/rotation.rs:26-37) for rot_n consecutive frames, applied to the
    // transforms before the matrices are derived (rot_n == 0: the model matrices are read as they are)
    sxe_transform *transforms;
    float *models_rw;
    float *models_snap;          // a copy of the frame's model matrices kept with the frame (SXE_SHADE_LIT), or null
    uint64_t rot_seed;
    uint32_t rot_first, rot_n;
    float rot_delta_ms;
};

// Per-tile bookkeeping of a frame.  Every tile owns `stride` consecutive entries of the bin array,
// cut into five regions whose capacities are the same for every tile: one per pixel-box size class,
// filled by k_setup itself (a wave reserves positions with one atomic per (tile, class) group of
// its triangles and stores the slot numbers there), and one for the late entries (triangles
// touching more than four tiles, fans of clipped triangles) that k_finish places.  Keeping the
// classes apart lets the lanes of a wave of the tile kernel walk boxes of similar size.  A frame
// that needs more room in some region than its capacity reports an overflow and is drawn again
// with larger regions.
constexpr int kSizeClasses = 4;
constexpr int kBinRegions = kSizeClasses + 1;
struct BinLayout {
    uint32_t cap[kBinRegions]; // entries a tile can hold per region (classes 0-3, late)
    uint32_t off[kBinRegions]; // where the region starts inside the tile's stride
    uint32_t stride;
};
// cap[r] / off[r] for a region number that is not known at compile time, by selects (indexing a
// kernel argument with a register would put the whole structure into scratch memory)
__host__ __device__ inline uint32_t pick_region(const uint32_t (&v)[kBinRegions], uint32_t r)
{
    return r == 0u ? v[0] : r == 1u ? v[1] : r == 2u ? v[2] : r == 3u ? v[3] : v[4];
}
struct __attribute__((aligned(16))) TileCtr {
    uint32_t cnt[kSizeClasses];
    uint32_t multi;
    uint32_t flags; // kTileBlend
    uint32_t pad1, pad2;
};
constexpr size_t kTileCtrWords = sizeof(TileCtr) / 4;
__host__ __device__ inline uint32_t bin_reserved(const TileCtr &t) { return t.cnt[0] + t.cnt[1] + t.cnt[2] + t.cnt[3]; }
__host__ __device__ inline uint32_t bin_size(const TileCtr &t) { return bin_reserved(t) + t.multi; }
// Size class of a triangle from the number of pixels in its box: <= 2, <= 8, <= 24, more
// (triangles too large for the 32-bit edge functions always count as the last class).
__host__ __device__ inline uint32_t size_class(uint32_t box_pixels, bool is_narrow)
{
    return !is_narrow ? 3u : box_pixels <= 2u ? 0u : box_pixels <= 8u ? 1u : box_pixels <= 24u ? 2u : 3u;
}

struct SetupArgs {
    const sxe_vertex *vertices;
    const uint32_t *indices;
    const MeshDev *meshes;
    const InstDev *insts;
    const BlockDesc *blocks;
    const float *pvm;          // 16 floats per object (caller order), per view
    const float *models;       // 16 floats per object (caller order): read by SXE_SHADE_LIT only
    LightDev light;
    const uint8_t *visible;
    uint32_t n_objects;
    uint32_t n_blocks;         // per view
    uint32_t shading;
    uint32_t debug;            // kDebug* switches (development only)
    FrameGeom g;
    uint32_t *blk_count;       // survivors per wave of every work item (four counts each)
    TileCtr *tiles;
    uint32_t *bins;
    BinLayout bl;
    const TexDev *textures;    // read only when blend_scene is set
    uint32_t blend_scene;      // some texture of the table is translucent
    RastRec *rast;
    WideRec *wide;
    ShadeRec *shade;
    ColorRec *color;
    uint32_t fan_capacity;     // records available past slot 256 * n_blocks; also the clip list's length
    uint32_t clip_off;         // the clip kernel does not follow: a triangle that needs it voids the frame
    ClipItem *clip_list;
    uint32_t *fan_order;       // per fan record (slot - 256 * n_blocks): the order slot of its primitive
    uint32_t *walk_list;       // slots of the triangles over more than four tiles and of the fan triangles
    FrameCounters *counters;
};

// What a tile's workgroup needs to start, one record per launch position (dearest tiles first).
struct __attribute__((aligned(16))) TileLaunch {
    uint32_t tile;                 // tile number inside its view
    uint32_t late, c3, c2, c1, c0; // entries per region, in the order the tile kernel takes them
    uint32_t view;
    uint32_t flags;                // the tile's TileCtr::flags
};

struct FinishArgs {
    FrameGeom g;
    const uint32_t *blk_count;
    uint32_t n_blocks;
    uint32_t walk_blocks;      // workgroups launched for the late entries (walk_list)
    const RastRec *rast;
    const WideRec *wide;
    const uint32_t *walk_list; // records whose triangles take their entries here, counters->walk_count of them
    const uint32_t *fan_order; // per fan record: the order slot of its primitive (which tells its view)
    const TileCtr *tiles;      // counts as k_setup and k_clip left them
    uint32_t *late_fill;       // per tile: how many of its late entries have been placed
    TileLaunch *tile_launch;   // out
    uint32_t *bins;
    BinLayout bl;
    FrameCounters *counters;
};

struct RasterArgs {
    FrameGeom g;
    const TileLaunch *tile_launch; // workgroup b renders tile tile_launch[b].tile
    const uint32_t *bins;
    BinLayout bl;
    const RastRec *rast;
    const WideRec *wide;
    const ShadeRec *shade;
    const ColorRec *color;
    const uint32_t *fan_order;  // per record past the first n_slots: the slot that ranks it
    uint32_t n_slots;           // 256 * setup work items * views
    size_t image_stride;        // pixels from one view's image to the next (views render into consecutive images)
    const FrameCounters *counters;
    FrameCounters *host_counters; // host-visible copy of the counters, written by the first workgroup (may be null)
    const uint2 *footprints; // the textures as tables of cells, two of which make a bilinear footprint (TexDev::offset indexes it)
    const TexDev *textures;
    uint32_t n_textures;
    uint8_t *out_color;   // RGBA8, row-major, full frame
    float *out_depth;     // may be null (SXE_FLAG_DISCARD_DEPTH)
    uint32_t *out_prim;   // may be null
    unsigned long long *tile_times; // development: 8 clock values per tile (phase stamps), or null
    uint32_t clear_code;  // encoded clear colour, little-endian RGBA
    uint32_t shading;
    uint32_t debug;
    uint32_t blend_scene; // tiles flagged kTileBlend are left to k_raster_blend
    uint32_t bgra;        // the colour image is stored in B, G, R, A byte order
};

constexpr int kMaxDevices = 64; // device ordinals the library keeps per-device state for
// Constant tables of the tile kernel (sRGB decode and encode): once per device, from sxe_create.
hipError_t init_device_tables(int device);

// Launchers (defined in sxe_geometry.hip and sxe_raster.hip). All enqueue on `s` and return the launch status.
hipError_t launch_object(const ObjectArgs &a, hipStream_t s);
hipError_t launch_transform_to_model(const sxe_transform *t, float *models, uint32_t n, hipStream_t s);
hipError_t launch_rotation_system(sxe_transform *t, float *models, uint32_t n, uint64_t seed,
                                  uint32_t frame, uint32_t n_frames, float delta_ms, hipStream_t s);
hipError_t launch_vertex(const float *px, const float *py, const float *pz, const MeshDev *meshes,
                         const VBlockDesc *vblocks, uint32_t n_vblocks, const float *pvm,
                         float4 *clip, hipStream_t s);
hipError_t launch_setup(const SetupArgs &a, hipStream_t s);
hipError_t launch_clip(const SetupArgs &a, hipStream_t s);
hipError_t launch_finish(const FinishArgs &a, hipStream_t s);
hipError_t launch_raster_shade(const RasterArgs &a, hipStream_t s);
hipError_t launch_raster_blend(const RasterArgs &a, hipStream_t s);
hipError_t launch_texture_footprints(const uint8_t *texels, uint32_t width, uint32_t height, uint2 *cells, hipStream_t s);

} // namespace sxe
