// sxe_internal.h -- device-side data layout and kernel launch interface of libsxe_hip.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sxe.h"
#include "sxe_math.h"

namespace sxe {

constexpr int kTile = 32;          // screen tile edge, pixels: one workgroup rasterises one tile
constexpr int kTilePixels = kTile * kTile;
constexpr int kSetupThreads = 256; // triangles per setup workgroup
constexpr int kRasterThreads = 256;
constexpr float kGuard = 4.0f;     // guard band: x,y are clipped at +-kGuard*w (power of two)
constexpr int kMaxFan = 7;         // a triangle clipped by six planes has at most 9 corners
constexpr unsigned long long kVisEmpty = ~0ull;
// development-only switches carried in sxe_config.flags (timing experiments; outputs are wrong)
constexpr uint32_t kDebugSkipRaster = 1u << 30, kDebugSkipResolve = 1u << 31;
constexpr uint32_t kDebugSkipSmall = 1u << 28, kDebugSkipLarge = 1u << 29, kDebugSkipAtomic = 1u << 27;
constexpr uint32_t kDebugMask = 31u << 27;

struct MeshDev {
    uint32_t vtx_base, n_vtx, idx_base, n_tri;
};

// One drawn object in draw order (texture groups ascending, caller order inside a group;
// src/renderer/drawer.rs:315-347,481-487).
struct InstDev {
    uint32_t mesh;
    uint32_t object;   // caller-order index: selects the model matrix
    int32_t texture;
    uint32_t prim_base; // id of this object's first triangle in the frame
};

struct BlockDesc { // work item of the setup kernels: 256 consecutive triangles of one object
    uint32_t inst, tri_start;
};

struct VBlockDesc { // work item of the vertex kernel: up to 1024 consecutive vertices of one object
    uint32_t object, vtx_start, mesh, clip_base;
};

struct TexDev {
    uint32_t offset; // first texel in the pooled texel buffer
    uint32_t width, height, pad;
};

// Everything the coverage/depth loop needs about one screen triangle (48 B, three 16-B loads).
struct __attribute__((aligned(16))) RastRec {
    int32_t x0, y0, x1, y1;
    int32_t x2, y2;
    float z0, dz1;
    float dz2, rA;
    uint32_t prim;
    int32_t texture;
};

// Perspective-correct attribute planes of the same triangle, read once per visible pixel.
struct __attribute__((aligned(16))) ShadeRec {
    float iw0, diw1, diw2, uw0;
    float duw1, duw2, vw0, dvw1;
    float dvw2, pad0, pad1, pad2;
};

struct __attribute__((aligned(16))) ColorRec {
    float rw0, drw1, drw2, gw0;
    float dgw1, dgw2, bw0, dbw1;
    float dbw2, pad0, pad1, pad2;
};

struct FrameCounters {
    uint32_t total_records, total_pairs, overflow, pad;
};

struct FrameGeom {
    uint32_t width, height;
    uint32_t sy0, sy1;       // scissor rows [sy0, sy1)
    uint32_t tiles_x, tile_row0, n_tile_rows, n_tiles;
    float half_w, half_h;
};

struct SetupArgs {
    const sxe_vertex *vertices;
    const uint32_t *indices;
    const MeshDev *meshes;
    const InstDev *insts;
    const BlockDesc *blocks;
    const float *pvm; // 16 floats per object (caller order)
    uint32_t n_blocks;
    uint32_t shading;
    FrameGeom g;
    uint32_t *blk_count;
    const uint32_t *blk_base;
    uint32_t *tile_count;
    const uint32_t *tile_off;
    uint32_t *tile_fill;
    RastRec *rast;
    ShadeRec *shade;
    ColorRec *color;
    uint32_t *bins;
    FrameCounters *counters;
};

struct ScanArgs {
    const uint32_t *blk_count;
    uint32_t *blk_base;
    uint32_t n_blocks;
    const uint32_t *tile_count;
    uint32_t *tile_off;
    uint32_t n_tiles;
    uint32_t rec_capacity, bin_capacity;
    FrameCounters *counters;
};

struct RasterArgs {
    FrameGeom g;
    const uint32_t *tile_count;
    const uint32_t *tile_off;
    const uint32_t *bins;
    const RastRec *rast;
    const ShadeRec *shade;
    const ColorRec *color;
    const FrameCounters *counters;
    const uint8_t *texels;
    const TexDev *textures;
    uint8_t *out_color;   // RGBA8, row-major, full frame
    float *out_depth;
    uint32_t *out_prim;   // may be null
    uint32_t clear_code;  // encoded clear colour, little-endian RGBA
    uint32_t shading;
    uint32_t debug;
};

// Launchers (defined in sxe_kernels.hip). All enqueue on `s` and return the launch status.
hipError_t launch_object(const float *models, const Mat4 &pv, float *pvm, uint32_t n, hipStream_t s);
hipError_t launch_transform_to_model(const sxe_transform *t, float *models, uint32_t n, hipStream_t s);
hipError_t launch_rotation_system(sxe_transform *t, float *models, uint32_t n, uint64_t seed,
                                  uint32_t frame, float delta_ms, hipStream_t s);
hipError_t launch_vertex(const float *px, const float *py, const float *pz, const MeshDev *meshes,
                         const VBlockDesc *vblocks, uint32_t n_vblocks, const float *pvm,
                         float4 *clip, hipStream_t s);
hipError_t launch_setup_count(const SetupArgs &a, hipStream_t s);
hipError_t launch_scan(const ScanArgs &a, hipStream_t s);
hipError_t launch_setup_emit(const SetupArgs &a, hipStream_t s);
hipError_t launch_raster_shade(const RasterArgs &a, hipStream_t s);

} // namespace sxe
