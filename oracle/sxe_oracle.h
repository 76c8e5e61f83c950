/*
 * sxe_oracle.h -- CPU oracle for the raster/shade path.  TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * A plain-C, single-threaded, immediate-mode restatement of what the reference renderer asks
 * its Vulkan driver to do for one frame, plus the reference's own matrix maths.  It exists so
 * that the HIP library can be compared with it; only tests, the smoke check and the benchmark's
 * CPU-baseline leg may build, load or call it.  The product library never links it.
 *
 * PARITY STATUS: "parity unpinned".  The reference (matthew-russo/renderer) has no tests, no
 * golden images and cannot be built offline, and its arithmetic lives in cgmath 0.17.0, gfx-hal
 * 0.5.0 and the Vulkan driver, none of which are vendored.  The oracle is pinned by closed-form
 * known answers derived from the cited reference formulas (see tests/test_oracle_math.py) and
 * by rasterization invariants, not by pixels the reference produced.
 *
 * Reference citations are relative to the reference repository root.
 */
#ifndef SXE_ORACLE_H
#define SXE_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/components/transform.rs:16-21; rotation in cgmath memory order (s, x, y, z). */
typedef struct {
    float position[3];
    float scale[3];
    float rotation[4];
} orc_transform;

/* src/primitives/vertex.rs:2-6; attribute offsets 0/12/24, stride 32 (src/renderer/drawer.rs:665-696). */
typedef struct {
    float pos[3];
    float color[3];
    float uv[2];
} orc_vertex;

typedef struct {
    uint32_t first_vertex, n_vertices;
    uint32_t first_index, n_indices; /* indices are mesh-local (src/components/mesh.rs:7) */
} orc_mesh;

/* One drawable in caller order (src/primitives/drawable.rs:7-12); its model matrix is
 * models[position in the array] (src/renderer/drawer.rs:460-465). */
typedef struct {
    uint32_t mesh;
    int32_t texture;
    uint32_t rendered; /* src/components/mesh.rs:8, skipped at src/renderer/drawer.rs:327-329 */
} orc_object;

typedef struct {
    const uint8_t *texels; /* RGBA8, sRGB-encoded colour, row-major, tightly packed */
    uint32_t width, height;
} orc_texture;

/* ORC_SHADE_LIT has no counterpart in the reference (standard.frag has no lighting, Vertex has no
 * normal): it restates the flat Blinn-Phong extension that include/sxe.h defines at sxe_set_light,
 * so that the kernels have something to be compared with.  Parity unpinned, like everything here. */
enum { ORC_SHADE_TEXTURE = 0, ORC_SHADE_VERTEX_COLOR = 1, ORC_SHADE_COLOR_TIMES_TEXTURE = 2, ORC_SHADE_LIT = 3 };
/* ORC_FLAG_REFERENCE_INDEXING: the draws as the reference binary records them, indexing flaws
 * included (src/renderer/drawer.rs:313-347; see draw_list in sxe_oracle.c).  Every object, rendered
 * or not, then needs a valid texture index. */
/* ORC_FLAG_BGRA: the colour target in B, G, R, A byte order (Bgra8Srgb, the first sRGB surface
 * format on X11, src/renderer/presenter.rs:472-478); the default is R, G, B, A (Rgba8Srgb). */
enum { ORC_FLAG_NO_TRIVIAL_REJECT = 1, ORC_FLAG_REFERENCE_INDEXING = 2, ORC_FLAG_BGRA = 4 };

typedef struct {
    uint32_t width, height;
    float clear[4];                 /* linear RGBA; src/renderer/drawer.rs:307 */
    uint32_t scissor_y0, scissor_y1; /* rows [y0, y1) rendered; full frame = [0, height) */
    uint32_t shading;
    uint32_t flags;
    /* ORC_SHADE_LIT only: directions in world space (any length > 0), terms, exponent 2^shininess_log2 */
    float to_light[3], to_viewer[3];
    float ambient, diffuse, specular;
    uint32_t shininess_log2;
} orc_config;

typedef struct {
    uint64_t prims, prims_rejected, prims_clipped, subtris, subtris_culled;
    uint64_t fragments_tested, fragments_covered, fragments_passed;
    uint64_t subtris_empty; /* set up and scanned, but no sample point inside */
} orc_stats;

/* ---- deterministic elementary functions (the numeric contract both sides implement) ---- */
void orc_sincosf(float x, float *s, float *c);
float orc_atan2f(float y, float x);
float orc_asinf(float x);

/* ---- reference matrix maths ---- */
void orc_quat_from_euler_deg(float x, float y, float z, float q[4]);
void orc_quat_mul(const float a[4], const float b[4], float out[4]);
void orc_quat_to_euler(const float q[4], float e[3]);
void orc_transform_identity(orc_transform *t);
void orc_transform_rotate(orc_transform *t, float x, float y, float z);
void orc_transform_translate(orc_transform *t, float x, float y, float z);
void orc_transform_to_model(const orc_transform *t, float m[16]);
void orc_mat4_mul(const float a[16], const float b[16], float out[16]);
void orc_fps_view(const float eye[3], float pitch, float yaw, float view[16]);
void orc_perspective_yflip(float fovy_deg, float aspect, float z_near, float z_far, float proj[16]);
void orc_camera(const orc_transform *cam, float fovy_deg, float aspect, float z_near, float z_far,
                float view[16], float proj[16]);
/* Deterministic per-frame rotation system (analogue of src/systems/rotation.rs:26-37). */
float orc_rand01(uint64_t seed, uint32_t frame, uint32_t index);
void orc_rotation_system(orc_transform *t, uint32_t n, uint64_t seed, uint32_t frame, float delta_ms);

/* ---- colour helpers ---- */
uint8_t orc_srgb_encode(float linear);
float orc_srgb_decode(uint8_t code);

/* ---- the frame ---- */
int orc_render(const orc_config *cfg, const orc_vertex *vertices, const uint32_t *indices,
               const orc_mesh *meshes, uint32_t n_meshes, const orc_object *objects,
               uint32_t n_objects, const float *models, const float view[16],
               const float proj[16], const orc_texture *textures, uint32_t n_textures,
               uint8_t *out_rgba, float *out_depth, uint32_t *out_prim, float *out_clip,
               orc_stats *stats);

/* Band-parallel variant used only as the benchmark's CPU baseline: the frame is cut into
 * `n_threads` horizontal bands rendered concurrently with the same per-band code. */
int orc_render_mt(const orc_config *cfg, const orc_vertex *vertices, const uint32_t *indices,
                  const orc_mesh *meshes, uint32_t n_meshes, const orc_object *objects,
                  uint32_t n_objects, const float *models, const float view[16],
                  const float proj[16], const orc_texture *textures, uint32_t n_textures,
                  uint8_t *out_rgba, float *out_depth, uint32_t *out_prim, int n_threads);

#ifdef __cplusplus
}
#endif
#endif
