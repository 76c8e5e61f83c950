/*
 * sxe.h -- C ABI of libsxe_hip.so, the MI355X (gfx950) raster/shade path.
 *
 * The reference renderer (matthew-russo/renderer) has no FFI: its seam is the crate-private Rust
 * trait `Drawer<B>` (src/renderer/drawer.rs:26-31) plus `Presenter<B>` (src/renderer/presenter.rs:16-22).
 * Every entry point below names the reference interface it stands in for.  All pointer arguments
 * are borrowed for the duration of the call only (the reference takes its arguments by value,
 * src/renderer/drawer.rs:458,496,507); the library copies what it keeps.
 *
 * Threading: a context is externally synchronised (one caller at a time, like the reference's
 * single engine thread, src/main.rs:143); distinct contexts are independent.  Each context owns
 * one HIP stream on one device.
 *
 * Errors: every function returns 0 or a negative SXE_E_* code and never aborts the process
 * (the reference panics instead: src/renderer/drawer.rs:453, src/renderer/types.rs:51).
 *
 * Values that are not finite (NaN, infinities) among vertex positions, attributes, matrices, poses
 * or the camera are no error: a triangle with a corner that cannot be projected is not drawn, one
 * that reaches beyond the clip volume is clipped, a colour that is a NaN is stored as 0, and the
 * rest of the frame is what it would be without them.  Index and texture numbers out of range are
 * refused when the scene is uploaded (SXE_E_INVALID).
 */
#ifndef SXE_H
#define SXE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SXE_OK 0
#define SXE_E_INVALID (-1)  /* bad argument or scene */
#define SXE_E_DEVICE (-2)   /* a HIP call failed; see sxe_last_error */
#define SXE_E_CAPACITY (-3) /* a stated limit was exceeded */
#define SXE_E_STATE (-4)    /* call order violated (e.g. draw before update_drawables) */

/* fragment shading variants: standard.frag:12 (default), standard.frag:13 (commented
 * alternative), ui.frag:12 */
#define SXE_SHADE_TEXTURE 0u
#define SXE_SHADE_VERTEX_COLOR 1u
#define SXE_SHADE_COLOR_TIMES_TEXTURE 2u
/* An extension without a counterpart in the reference, whose shaders have no lighting and whose
 * Vertex has no normal: flat Blinn-Phong over the texture, see sxe_set_light. */
#define SXE_SHADE_LIT 3u

#define SXE_FLAG_WRITE_COVERAGE 1u /* keep a per-pixel primitive-id buffer (parity tests) */
#define SXE_FLAG_TIMING 2u         /* bracket every kernel with HIP events (sxe_get_timings) */
/* Build the draw list exactly as the reference binary records its command buffer
 * (src/renderer/drawer.rs:313-347), three indexing flaws included:
 *   1. inside a texture group the i-th mesh (unrendered ones counted) is drawn with the i-th model
 *      matrix of the whole scene (dynamic offset i * stride, :326,331), not with its own;
 *   2. its index range starts where the ranges of the draws recorded before it, in group order,
 *      end (:313,345-346), although the index buffer is laid out in drawable order (:181-193);
 *   3. an unrendered mesh does not move that cursor (:327-329).
 * Scenes with one texture group and every mesh rendered come out the same either way.  Without
 * the flag (the default) every rendered object draws its own mesh under its own matrix, which is
 * what the reference means; with it, the images are the ones the reference binary produces for
 * the same drawables, e.g. for its own demo world of 64 cubes over 3 textures (src/main.rs:201-227).
 * Every object then needs a valid texture index, rendered or not (it takes a place in its group).
 * sxe_read_clip_positions is not affected: it reports every drawable's vertices under its own matrix. */
#define SXE_FLAG_COMPAT_REFERENCE_INDEXING 4u
/* Store the colour images in B, G, R, A byte order.  The reference renders into the first sRGB
 * format its surface offers (src/renderer/presenter.rs:472-478), which on X11 is Bgra8Srgb; the
 * default here is R, G, B, A (its fallback, Rgba8Srgb).  Same values, other byte order. */
#define SXE_FLAG_BGRA 8u
/* Do not keep the depth images.  The reference's depth attachment is cleared when the pass begins
 * and its contents are not stored when it ends (AttachmentStoreOp::DontCare,
 * src/renderer/drawer.rs:544-547): depth exists for the test inside the frame only.  By default the
 * library stores every frame's depth image for sxe_read_depth (what the parity tests compare); with
 * this flag the depth test is the same, nothing is written, the images are not allocated and
 * sxe_read_depth returns SXE_E_STATE. */
#define SXE_FLAG_DISCARD_DEPTH 16u

#define SXE_MAX_DIM 8192u          /* width and height limit (8K frames fit) */
#define SXE_MAX_PRIMS (1u << 29)   /* triangles per frame */
#define SXE_NOALLOC_SCRATCH_BYTES (32u << 20) /* static work memory of sxe_draw_noalloc, per GPU */

typedef struct sxe_ctx sxe_ctx;

/* src/primitives/vertex.rs:2-6; offsets 0/12/24, stride 32 as hard-coded at
 * src/renderer/drawer.rs:665-696. */
typedef struct {
    float pos[3];
    float color[3];
    float uv[2];
} sxe_vertex;

/* src/components/transform.rs:16-21; rotation is a quaternion in cgmath memory order s,x,y,z. */
typedef struct {
    float position[3];
    float scale[3];
    float rotation[4];
} sxe_transform;

/* What GfxDrawer::new receives (src/renderer/drawer.rs:55) plus the constants it hard-codes:
 * viewport extent (src/renderer/presenter.rs:373-383), perspective parameters
 * (src/renderer/drawer.rs:248-253), clear colour (:307). */
typedef struct {
    uint32_t width, height;
    float fov_deg, z_near, z_far;
    float clear[4];
    int32_t device;                  /* HIP device ordinal */
    uint32_t n_images;               /* swapchain images (src/renderer/presenter.rs:440) */
    uint32_t scissor_y0, scissor_y1; /* rows [y0,y1) this context renders (sort-first shard);
                                        0,0 means the whole frame */
    uint32_t shading;                /* SXE_SHADE_* */
    uint32_t flags;                  /* SXE_FLAG_* */
} sxe_config;

/* RGBA8 texels, sRGB-encoded colour, tightly packed rows: what TextureData::load produces
 * (src/renderer/types.rs:158-188) after JPEG decoding, which is outside this library. */
typedef struct {
    const uint8_t *rgba;
    uint32_t width, height;
} sxe_texture_desc;

/* src/primitives/drawable.rs:7-12 with the mesh inlined (src/components/mesh.rs:4-9). */
typedef struct {
    const sxe_vertex *vertices;
    uint32_t n_vertices;
    const uint32_t *indices; /* mesh-local, triangle list */
    uint32_t n_indices;
    sxe_transform transform;
    int32_t texture;   /* index into the table given to sxe_set_textures */
    uint32_t rendered; /* src/components/mesh.rs:8 */
} sxe_drawable;

typedef struct {
    const sxe_vertex *vertices;
    uint32_t n_vertices;
    const uint32_t *indices;
    uint32_t n_indices;
} sxe_mesh_desc;

typedef struct {
    uint32_t mesh;
    int32_t texture;
    uint32_t rendered;
    sxe_transform transform;
} sxe_instance;

/* Per-kernel device times of the last synchronised frame, milliseconds (needs SXE_FLAG_TIMING).
 * A rotation-system update is applied by the frame's first kernel and counted in object_ms;
 * transform_ms is 0. */
typedef struct {
    float transform_ms, object_ms, setup_ms, finish_ms, raster_shade_ms, frame_ms;
} sxe_timings;

/* Work counters of the last synchronised frame. */
typedef struct {
    uint64_t prims;        /* triangles submitted */
    uint64_t records;      /* triangles that survived setup */
    uint64_t bin_entries;  /* (triangle, tile) pairs */
    uint64_t tiles;        /* screen tiles rendered */
    uint64_t instance_vertices;
    uint64_t objects;
    uint32_t tile_size;
    uint32_t redraws;      /* frames re-rendered after growing a work buffer */
} sxe_frame_stats;

/* ---- lifetime: GfxDrawer::new / Drop (src/renderer/drawer.rs:55-160,360-388) ---- */
void sxe_default_config(sxe_config *cfg, uint32_t width, uint32_t height);
int sxe_create(const sxe_config *cfg, sxe_ctx **out);
void sxe_destroy(sxe_ctx *ctx);
/* Message of the last failure on this context (or of the last failed sxe_create when ctx is NULL). */
const char *sxe_last_error(const sxe_ctx *ctx);
const char *sxe_version(void);

/* ---- scene: Drawer::update_drawables (src/renderer/drawer.rs:458-494) ---- */
/* generate_images (src/renderer/drawer.rs:216-236): replaces the whole texture table.  Opaque
 * textures (alpha 255 everywhere, as every JPEG-derived reference texture is) are shaded once per
 * pixel.  A texture with any other alpha makes the screen tiles its triangles touch take an ordered
 * path that depth-tests and blends every fragment in submission order (BlendState::ALPHA,
 * src/renderer/drawer.rs:660-663): exact and much slower (a 32x32 tile that holds more than 4096
 * such triangles is sorted and blended in consecutive stretches of the submission order).
 * On the device a texture is kept as a table of (width + 2) x (height + 1) cells of 8 bytes (a texel
 * and the one below it, clamp to edge applied; two neighbouring cells are the four texels of a
 * sample) next to the texels as given: about 12 bytes of device memory per texel; all tables
 * together hold fewer than 2^32 cells (SXE_E_CAPACITY). */
int sxe_set_textures(sxe_ctx *ctx, const sxe_texture_desc *textures, uint32_t n);
/* One mesh per drawable, drawn once, as the reference stores them (src/renderer/drawer.rs:173-193). */
int sxe_update_drawables(sxe_ctx *ctx, const sxe_drawable *drawables, uint32_t n);
/* Same scene contract with shared meshes; lifts the reference's 1024-object cap
 * (src/renderer/allocator.rs:311-318). */
int sxe_update_instances(sxe_ctx *ctx, const sxe_mesh_desc *meshes, uint32_t n_meshes,
                         const sxe_instance *instances, uint32_t n_instances);

/* ---- per frame ---- */
/* Drawer::update_uniforms (src/renderer/drawer.rs:496-505): n column-major mat4, caller order. */
int sxe_update_uniforms(sxe_ctx *ctx, const float *models, uint32_t n);
/* fetch_uniforms' Transform::to_ubo loop (src/main.rs:237-244, src/components/transform.rs:69-80)
 * run on the device: uploads n transforms and derives the model matrices there. */
int sxe_update_transforms(sxe_ctx *ctx, const sxe_transform *transforms, uint32_t n);
/* Rotation::run (src/systems/rotation.rs:26-37) on the device-resident transforms, with a
 * counter-based PRNG in place of thread_rng and a caller-supplied frame delta in ms.  The update
 * is carried out by the first kernel of the next sxe_draw (or at once by whatever call reads or
 * writes transforms or model matrices before that). */
int sxe_rotation_system(sxe_ctx *ctx, uint64_t seed, uint32_t frame, float delta_ms);
/* The same for n_frames consecutive frames in one launch (identical arithmetic, applied in frame
 * order): what a process that renders only every N-th frame of an animation needs. */
int sxe_rotation_system_span(sxe_ctx *ctx, uint64_t seed, uint32_t first_frame, uint32_t n_frames, float delta_ms);
/* Drawer::update_camera (src/renderer/drawer.rs:507-518). */
int sxe_update_camera(sxe_ctx *ctx, const sxe_transform *camera);
/* The same upload with caller-built matrices (column-major view, proj). */
int sxe_update_camera_matrices(sxe_ctx *ctx, const float *view, const float *proj);
/* Drawer::draw (src/renderer/drawer.rs:412-456): enqueue one frame into image `image_index`;
 * asynchronous on the context's stream. */
int sxe_draw(sxe_ctx *ctx, uint32_t image_index);
/* update_camera + draw (src/renderer/drawer.rs:507-518, :412-456) for n independent views of the
 * resident scene in one go: view v is seen by cameras[v] and rendered into image first_image + v
 * (the context needs that many images).  The views of a set share every kernel launch (the view is
 * one more grid dimension), which is what makes thousands of small views cheap: one at a time they
 * cost a launch chain each.  Every view comes out exactly as sxe_update_camera + sxe_draw would
 * render it.  The library cuts n into sets that fit its work buffers (up to 64 views) and enqueues
 * them like frames of sxe_draw; a pending rotation-system update is applied first, once. */
int sxe_draw_views(sxe_ctx *ctx, const sxe_transform *cameras, uint32_t n, uint32_t first_image);
/* The fence wait of src/renderer/drawer.rs:416-424.  Reports deferred device-side failures. */
int sxe_sync(sxe_ctx *ctx);
/* sxe_draw + sxe_sync for callers that cannot allocate (tight serving loops with a fixed memory
 * budget, harnesses that forbid hipMalloc between frames).  No counterpart in the reference.  Same
 * inputs (the context's scene, uniforms and camera) and same output (image `image_index`) as
 * sxe_draw, but every work buffer of the frame (per-object matrices, triangle records, bin tables,
 * bins) lives in a static device array of SXE_NOALLOC_SCRATCH_BYTES that is part of the library's
 * code object; the call makes no hipMalloc/hipFree and returns when the frame is finished (it waits
 * on the host for the frame's event, so it cannot itself be recorded into a HIP graph).
 * Limits (SXE_E_CAPACITY instead of growing anything):
 *   - the array must hold about 100 bytes per submitted triangle (148 with a vertex-colour shading
 *     mode; objects count in steps of 256 triangles) and 72 bytes per screen tile;
 *   - what is left is divided evenly among the tiles for their bins (4 bytes per entry), at least 768
 *     entries each: frames of more than about 10,000 tiles (any 8K frame) do not fit whatever the
 *     scene, and a frame whose fullest tile needs more than its share fails as well;
 *   - at most 4096 triangles of a frame may cross the near, far or a guard-band plane, and their
 *     clipped pieces may number at most 4096.
 * Room for roughly 200,000 triangles at 1080p.  No frame of sxe_draw may be in flight on the
 * context (SXE_E_STATE; call sxe_sync first).  Every call on one GPU uses that GPU's one array, from
 * whatever context or thread: a call that finds another one in progress on its GPU returns
 * SXE_E_STATE at once and draws nothing. */
int sxe_draw_noalloc(sxe_ctx *ctx, uint32_t image_index);

/* ---- readback: stands in for Presenter::present (src/renderer/presenter.rs:415-430) ---- */
int sxe_read_color(sxe_ctx *ctx, uint32_t image_index, uint8_t *rgba, size_t size_bytes);
/* The same frame without a copy into caller memory: the library transfers it into pinned memory
 * of its own and hands out the address (width * height * 4 bytes, valid until the next
 * sxe_map_color of this image or sxe_destroy).  Several times the rate of sxe_read_color into
 * pageable memory; what a presenter that only forwards the pixels wants. */
int sxe_map_color(sxe_ctx *ctx, uint32_t image_index, const uint8_t **rgba);
int sxe_read_depth(sxe_ctx *ctx, uint32_t image_index, float *depth, size_t size_bytes);
int sxe_read_coverage(sxe_ctx *ctx, uint32_t image_index, uint32_t *prim_ids, size_t size_bytes);
/* gl_Position of every vertex (standard.vert:28) in the layout of the reference's concatenated
 * vertex buffer (drawables in caller order, src/renderer/drawer.rs:173-179); 4 floats each. */
int sxe_read_clip_positions(sxe_ctx *ctx, float *clip, size_t size_bytes);
int sxe_read_models(sxe_ctx *ctx, float *models, size_t size_bytes);
int sxe_read_transforms(sxe_ctx *ctx, sxe_transform *transforms, size_t size_bytes);
/* Device address of an image's RGBA8 rows (for a zero-copy RCCL tile gather). */
int sxe_color_device_ptr(sxe_ctx *ctx, uint32_t image_index, void **ptr);
/* The context's HIP stream (a hipStream_t), so that a consumer of finished frames can order its own
 * work after a draw with events instead of a host wait: the counterpart of the present semaphore
 * that Drawer::draw signals (src/renderer/drawer.rs:437-452). */
int sxe_stream(sxe_ctx *ctx, void **stream);
int sxe_get_timings(sxe_ctx *ctx, sxe_timings *out);
/* Sums of the per-kernel times over every frame whose result has been taken since the last
 * reset, and the number of those frames: lets a caller keep several frames in flight and still
 * average kernel times (no counterpart in the reference, which has no timers). */
int sxe_get_timing_totals(sxe_ctx *ctx, sxe_timings *sums, uint64_t *frames);
int sxe_reset_timing_totals(sxe_ctx *ctx);
/* Every event costs a few microseconds of stream time.  With a stride of n only every n-th frame
 * drawn from now on is bracketed by events (default 1: every frame; 0: no frame, until the stride
 * is set again); sxe_get_timings then reports the last such frame and the totals count only those
 * (no counterpart in the reference). */
int sxe_set_timing_stride(sxe_ctx *ctx, uint32_t every);
/* Diagnostics: bin occupancy of the last frame, one count per rendered tile in row-major tile
 * order (load-balance analysis; no counterpart in the reference). */
int sxe_read_tile_counts(sxe_ctx *ctx, uint32_t *counts, size_t size_bytes);
/* Diagnostics: the same occupancy split into its parts, five counts per tile: entries by pixel-box
 * size class (four classes, smallest boxes first), then entries of triangles spanning more than
 * four tiles (no counterpart in the reference). */
int sxe_read_tile_parts(sxe_ctx *ctx, uint32_t *parts, size_t size_bytes);
/* Diagnostics: device clock (100 MHz) stamps of every tile's workgroup in the last frame, eight
 * values per tile (0: start, 1: after initialisation, 2-4: coverage phases, 5: compaction,
 * 7: end; 6 unused; a tile drawn by the ordered blending kernel instead: 0: start, 4: entries
 * gathered, 1: sorted, 2: entries done, 3: end, and in place of a stamp 5: its number of entries);
 * only for contexts created with the development flag 1u << 26. */
int sxe_read_tile_times(sxe_ctx *ctx, uint64_t *clocks, size_t size_bytes);
int sxe_get_stats(sxe_ctx *ctx, sxe_frame_stats *out);

/* Light of SXE_SHADE_LIT (no counterpart in the reference).  One directional light and a viewer at
 * infinity, both directions given in world space and normalised by the library; every triangle
 * is lit as a whole with the normal of its face: n = normalize(cross(w1 - w0, w2 - w0)) of its
 * corners' world positions (model * position), turned towards the viewer when it faces away
 * (faces are never culled, src/renderer/drawer.rs:653-657, so both sides are seen), then
 *   a = ambient + diffuse * max(n . to_light, 0)
 *   s = specular * max(n . h, 0)^(2^shininess_log2),   h = normalize(to_light + to_viewer)
 *   rgb = texture.rgb * a + s,   alpha = texture.a
 * in fp32, every product and sum rounded on its own, the power by repeated squaring.  A triangle
 * without area in world space gets a = ambient, s = 0.  The defaults are the ones of
 * sxe_default_light. */
typedef struct sxe_light {
    float to_light[3];
    float to_viewer[3];
    float ambient, diffuse, specular;
    uint32_t shininess_log2; /* at most 16 */
} sxe_light;
void sxe_default_light(sxe_light *light);
/* Takes effect with the next frame drawn.  SXE_E_INVALID for a direction of length zero or one
 * that is not finite, and for shininess_log2 above 16. */
int sxe_set_light(sxe_ctx *ctx, const sxe_light *light);
/* The unit vectors the kernels use (host arithmetic, no device needed): towards the light, towards
 * the viewer, and the half vector between them.  Returns SXE_E_INVALID as sxe_set_light does. */
int sxe_light_vectors(const sxe_light *light, float to_light[3], float to_viewer[3], float half[3]);

/* ---- host-side Transform / camera maths (no device needed) ---- */
void sxe_transform_identity(sxe_transform *t);                           /* transform.rs:24-30 */
void sxe_transform_translate(sxe_transform *t, float x, float y, float z); /* transform.rs:33-35 */
void sxe_transform_rotate(sxe_transform *t, float x, float y, float z);  /* transform.rs:37-46 */
void sxe_transform_to_model(const sxe_transform *t, float model[16]);    /* transform.rs:69-80 */
/* update_camera_uniform_buffer_object (src/renderer/drawer.rs:242-258). */
void sxe_camera_matrices(const sxe_transform *camera, float fov_deg, float aspect, float z_near,
                         float z_far, float view[16], float proj[16]);

/* The linear -> sRGB8 encoder of the tile kernel (table and lookup), evaluated on the host: lets
 * a test check it against the threshold definition for any float without a device (no
 * counterpart in the reference, whose encode happens in the Vulkan driver's output merger). */
int sxe_srgb_encode_host(const float *x, uint8_t *codes, size_t n);

#ifdef __cplusplus
}
#endif
#endif
