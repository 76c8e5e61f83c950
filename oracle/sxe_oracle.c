/*
 * sxe_oracle.c -- CPU oracle for the raster/shade path.  TEST INFRASTRUCTURE, NOT PRODUCT.
 * See sxe_oracle.h for its role and parity status ("parity unpinned": the reference has no
 * tests or golden images; known answers derived from its formulas pin this file instead).
 *
 * Build with -ffp-contract=off and without fast-math: every float expression below is meant to
 * be evaluated exactly as written, one IEEE-754 binary32 rounding per operation, so that the
 * HIP kernels (built with the same contract) can be compared bit for bit.
 *
 * Matrices are column-major, m[4*col + row], as cgmath's Matrix4 and GLSL's mat4.
 */
#include "sxe_oracle.h"
#include "orc_srgb_tables.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* Deterministic elementary functions.                                                          */
/* The reference calls cgmath, which calls the platform libm (src/components/transform.rs:38-42, */
/* src/renderer/drawer.rs:244,391-395).  libm results differ in the last bit between hosts and   */
/* GPU maths libraries, so the numeric contract fixes these polynomial forms instead (classic    */
/* single-precision Cody-Waite reduction and minimax polynomials, about 1 ulp); the tests check  */
/* them against libm to 2e-7 absolute.                                                           */
/* ------------------------------------------------------------------------------------------ */

void orc_sincosf(float x, float *s, float *c)
{
    float ax = fabsf(x);
    if (!(ax <= 1.0e6f)) { /* huge or NaN: defined result, never used by sane scenes */
        *s = 0.0f;
        *c = 1.0f;
        return;
    }
    int32_t j = (int32_t)(ax * 1.27323954473516f); /* 4/pi */
    j = (j + 1) & ~1;                              /* nearest even octant */
    float y = (float)j;
    float r = ((ax - y * 0.78515625f) - y * 2.4187564849853515625e-4f) - y * 3.77489497744594108e-8f;
    float z = r * r;
    float ps = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    float pc = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z;
    pc = pc - 0.5f * z;
    pc = pc + 1.0f;
    float sv, cv;
    switch ((j >> 1) & 3) {
    case 0: sv = ps; cv = pc; break;
    case 1: sv = pc; cv = -ps; break;
    case 2: sv = -ps; cv = -pc; break;
    default: sv = -pc; cv = ps; break;
    }
    *s = (x < 0.0f) ? -sv : sv;
    *c = cv;
}

static float orc_atanf_pos(float x) /* x >= 0 */
{
    float y;
    if (x > 2.414213562373095f) {
        y = 1.5707963267948966f;
        x = -(1.0f / x);
    } else if (x > 0.4142135623730950f) {
        y = 0.7853981633974483f;
        x = (x - 1.0f) / (x + 1.0f);
    } else {
        y = 0.0f;
    }
    float z = x * x;
    float p = (((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * x + x;
    return y + p;
}

float orc_atan2f(float y, float x)
{
    const float PI = 3.14159265358979323846f;
    const float PIO2 = 1.5707963267948966f;
    if (x == 0.0f) {
        if (y < 0.0f) return -PIO2;
        if (y == 0.0f) return 0.0f;
        return PIO2;
    }
    if (y == 0.0f) return (x < 0.0f) ? PI : 0.0f;
    float q = y / x;
    float a = (q < 0.0f) ? -orc_atanf_pos(-q) : orc_atanf_pos(q);
    if (x < 0.0f) return (y < 0.0f) ? a - PI : a + PI;
    return a;
}

float orc_asinf(float x)
{
    float a = fabsf(x);
    if (!(a <= 1.0f)) return (x < 0.0f) ? -1.5707963267948966f : 1.5707963267948966f; /* clamp domain */
    if (a < 1.0e-4f) return x;
    float z, w;
    int flag = a > 0.5f;
    if (flag) {
        z = 0.5f * (1.0f - a);
        w = sqrtf(z);
    } else {
        w = a;
        z = w * w;
    }
    float p = ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z + 7.4953002686e-2f) * z + 1.6666752422e-1f) * z * w + w;
    if (flag) {
        p = p + p;
        p = 1.5707963267948966f - p;
    }
    return (x < 0.0f) ? -p : p;
}

/* ------------------------------------------------------------------------------------------ */
/* cgmath 0.17.0 formulas used by the reference.                                                */
/* ------------------------------------------------------------------------------------------ */

/* Quaternion::from(Euler{Deg x, Deg y, Deg z}); src/components/transform.rs:28,38-42.
 * Deg -> Rad multiplies by pi/180 in f32; half angles; q = (s, x, y, z). */
void orc_quat_from_euler_deg(float x, float y, float z, float q[4])
{
    const float D2R = 0.017453292519943295f;
    float sx, cx, sy, cy, sz, cz;
    orc_sincosf((x * D2R) * 0.5f, &sx, &cx);
    orc_sincosf((y * D2R) * 0.5f, &sy, &cy);
    orc_sincosf((z * D2R) * 0.5f, &sz, &cz);
    q[0] = -sx * sy * sz + cx * cy * cz;
    q[1] = sx * cy * cz + sy * sz * cx;
    q[2] = -sx * sz * cy + sy * cx * cz;
    q[3] = sx * sy * cz + sz * cx * cy;
}

/* Hamilton product a*b; src/components/transform.rs:44. */
void orc_quat_mul(const float a[4], const float b[4], float out[4])
{
    float aw = a[0], ax = a[1], ay = a[2], az = a[3];
    float bw = b[0], bx = b[1], by = b[2], bz = b[3];
    float w = aw * bw - ax * bx - ay * by - az * bz;
    float x = aw * bx + ax * bw + ay * bz - az * by;
    float y = aw * by + ay * bw + az * bx - ax * bz;
    float z = aw * bz + az * bw + ax * by - ay * bx;
    out[0] = w; out[1] = x; out[2] = y; out[3] = z;
}

/* Euler::from(Quaternion); src/renderer/drawer.rs:244. */
void orc_quat_to_euler(const float q[4], float e[3])
{
    float qw = q[0], qx = q[1], qy = q[2], qz = q[3];
    float sqw = qw * qw, sqx = qx * qx, sqy = qy * qy, sqz = qz * qz;
    float unit = sqx + sqz + sqy + sqw;
    float test = qx * qz + qy * qw;
    if (test > 0.499f * unit) {
        e[0] = 0.0f;
        e[1] = 1.5707963267948966f;
        e[2] = orc_atan2f(qx, qw) * 2.0f;
    } else if (test < -0.499f * unit) {
        e[0] = 0.0f;
        e[1] = -1.5707963267948966f;
        e[2] = -orc_atan2f(qx, qw) * 2.0f;
    } else {
        e[0] = orc_atan2f(2.0f * (-qy * qz + qx * qw), 1.0f - 2.0f * (sqx + sqy));
        e[1] = orc_asinf(2.0f * (qx * qz + qy * qw));
        e[2] = orc_atan2f(2.0f * (-qx * qy + qz * qw), 1.0f - 2.0f * (sqy + sqz));
    }
}

/* Transform::new(); src/components/transform.rs:24-30. */
void orc_transform_identity(orc_transform *t)
{
    t->position[0] = t->position[1] = t->position[2] = 0.0f;
    t->scale[0] = t->scale[1] = t->scale[2] = 1.0f;
    orc_quat_from_euler_deg(0.0f, 0.0f, 0.0f, t->rotation);
}

/* Transform::rotate; src/components/transform.rs:37-46 (right-multiply, no renormalisation). */
void orc_transform_rotate(orc_transform *t, float x, float y, float z)
{
    float by[4], r[4];
    orc_quat_from_euler_deg(x, y, z, by);
    orc_quat_mul(t->rotation, by, r);
    memcpy(t->rotation, r, sizeof r);
}

/* Transform::translate; src/components/transform.rs:33-35. */
void orc_transform_translate(orc_transform *t, float x, float y, float z)
{
    t->position[0] = t->position[0] + x;
    t->position[1] = t->position[1] + y;
    t->position[2] = t->position[2] + z;
}

/* Matrix4 * Matrix4: column j of the product is ((a.x*b[j].x + a.y*b[j].y) + a.z*b[j].z) + a.w*b[j].w
 * over a's columns, summed left to right. */
void orc_mat4_mul(const float a[16], const float b[16], float out[16])
{
    float r[16];
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i)
            r[4 * j + i] = ((a[0 + i] * b[4 * j + 0] + a[4 + i] * b[4 * j + 1]) + a[8 + i] * b[4 * j + 2]) + a[12 + i] * b[4 * j + 3];
    memcpy(out, r, sizeof r);
}

static void mat4_identity(float m[16])
{
    memset(m, 0, 16 * sizeof(float));
    m[0] = m[5] = m[10] = m[15] = 1.0f;
}

/* Matrix4::from(Quaternion); used at src/components/transform.rs:76. */
static void mat4_from_quat(const float q[4], float m[16])
{
    float s = q[0], x = q[1], y = q[2], z = q[3];
    float x2 = x + x, y2 = y + y, z2 = z + z;
    float xx2 = x2 * x, xy2 = x2 * y, xz2 = x2 * z;
    float yy2 = y2 * y, yz2 = y2 * z, zz2 = z2 * z;
    float sy2 = y2 * s, sz2 = z2 * s, sx2 = x2 * s;
    m[0] = 1.0f - yy2 - zz2; m[1] = xy2 + sz2;        m[2] = xz2 - sy2;         m[3] = 0.0f;
    m[4] = xy2 - sz2;        m[5] = 1.0f - xx2 - zz2; m[6] = yz2 + sx2;         m[7] = 0.0f;
    m[8] = xz2 + sy2;        m[9] = yz2 - sx2;        m[10] = 1.0f - xx2 - yy2; m[11] = 0.0f;
    m[12] = 0.0f;            m[13] = 0.0f;            m[14] = 0.0f;             m[15] = 1.0f;
}

/* Transform::to_ubo: identity.concat(T).concat(R).concat(S); src/components/transform.rs:69-80. */
void orc_transform_to_model(const orc_transform *t, float m[16])
{
    float I[16], T[16], R[16], S[16], a[16], b[16];
    mat4_identity(I);
    mat4_identity(T);
    T[12] = t->position[0]; T[13] = t->position[1]; T[14] = t->position[2];
    mat4_from_quat(t->rotation, R);
    mat4_identity(S);
    S[0] = t->scale[0]; S[5] = t->scale[1]; S[10] = t->scale[2];
    orc_mat4_mul(I, T, a);
    orc_mat4_mul(a, R, b);
    orc_mat4_mul(b, S, m);
}

/* fps_view_matrix; src/renderer/drawer.rs:390-409. */
void orc_fps_view(const float eye[3], float pitch, float yaw, float view[16])
{
    float sp, cp, sy, cy;
    orc_sincosf(pitch, &sp, &cp);
    orc_sincosf(yaw, &sy, &cy);
    float xa[3] = { cy, 0.0f, -sy };
    float ya[3] = { sy * sp, cp, cy * sp };
    float za[3] = { sy * cp, -sp, cp * cy };
    view[0] = xa[0]; view[1] = ya[0]; view[2] = za[0]; view[3] = 0.0f;
    view[4] = xa[1]; view[5] = ya[1]; view[6] = za[1]; view[7] = 0.0f;
    view[8] = xa[2]; view[9] = ya[2]; view[10] = za[2]; view[11] = 0.0f;
    view[12] = ((xa[0] * eye[0] + xa[1] * eye[1]) + xa[2] * eye[2]) * -1.0f;
    view[13] = ((ya[0] * eye[0] + ya[1] * eye[1]) + ya[2] * eye[2]) * -1.0f;
    view[14] = ((za[0] * eye[0] + za[1] * eye[1]) + za[2] * eye[2]) * -1.0f;
    view[15] = 1.0f;
}

/* cgmath::perspective(Deg(fovy), aspect, near, far) then proj.y.y *= -1; src/renderer/drawer.rs:248-255.
 * The matrix keeps OpenGL's z range; only Y is flipped. */
void orc_perspective_yflip(float fovy_deg, float aspect, float z_near, float z_far, float proj[16])
{
    float s, c;
    orc_sincosf((fovy_deg * 0.017453292519943295f) / 2.0f, &s, &c);
    float f = c / s; /* cot(fovy/2) */
    memset(proj, 0, 16 * sizeof(float));
    proj[0] = f / aspect;
    proj[5] = f * -1.0f;
    proj[10] = (z_far + z_near) / (z_near - z_far);
    proj[11] = -1.0f;
    proj[14] = (2.0f * z_far * z_near) / (z_near - z_far);
}

/* update_camera_uniform_buffer_object; src/renderer/drawer.rs:242-258.  Note the argument order
 * at :246: pitch = euler.y, yaw = euler.x; roll is ignored. */
void orc_camera(const orc_transform *cam, float fovy_deg, float aspect, float z_near, float z_far,
                float view[16], float proj[16])
{
    float e[3];
    orc_quat_to_euler(cam->rotation, e);
    orc_fps_view(cam->position, e[1], e[0], view);
    orc_perspective_yflip(fovy_deg, aspect, z_near, z_far, proj);
}

/* Deterministic stand-in for thread_rng().gen::<f32>() at src/systems/rotation.rs:33:
 * SplitMix64 finaliser over (seed, frame, index), top 24 bits -> [0,1). */
float orc_rand01(uint64_t seed, uint32_t frame, uint32_t index)
{
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * ((((uint64_t)frame) << 32 | (uint64_t)index) + 1ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (float)(uint32_t)(z >> 40) * 5.9604644775390625e-8f; /* 2^-24 */
}

/* Rotation::run; src/systems/rotation.rs:26-37 with the PRNG above and a caller-fixed delta. */
void orc_rotation_system(orc_transform *t, uint32_t n, uint64_t seed, uint32_t frame, float delta_ms)
{
    float delta_time = delta_ms * 0.01f;
    for (uint32_t i = 0; i < n; ++i) {
        float r = orc_rand01(seed, frame, i);
        float ud = r * delta_time * 2.0f;
        orc_transform_rotate(&t[i], 2.0f * ud, 10.0f * ud, 20.0f * ud);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* Colour.                                                                                      */
/* ------------------------------------------------------------------------------------------ */

float orc_srgb_decode(uint8_t code) { return ORC_SRGB_DECODE[code]; }

/* Number of thresholds <= x (NaN and negatives give 0). */
uint8_t orc_srgb_encode(float x)
{
    int lo = 0, hi = 255; /* invariant: thr[0..lo) <= x, thr[hi..255) > x */
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (ORC_SRGB_ENCODE_THR[mid] <= x) lo = mid + 1; else hi = mid;
    }
    return (uint8_t)lo;
}

static uint8_t unorm8_encode(float a)
{
    if (!(a > 0.0f)) a = 0.0f;
    if (a > 1.0f) a = 1.0f;
    return (uint8_t)(int)rintf(a * 255.0f);
}

/* ------------------------------------------------------------------------------------------ */
/* The frame: vertex stage, clip, setup, raster, depth, shade, blend.                           */
/* Pipeline state restated from src/renderer/drawer.rs:652-705 (triangle list, fill, no cull,   */
/* alpha blend, depth Less + write), :302-311 (clear), src/renderer/presenter.rs:373-383        */
/* (viewport 0..W x 0..H, depth 0..1), :472-478 (sRGB target).                                  */
/* ------------------------------------------------------------------------------------------ */

#define GUARD 4.0f         /* guard band: clip x,y at +-GUARD*w (power of two) */
#define SUBPIX 256         /* 8 sub-pixel bits, round-half-even snap */
#define SNAP_LIMIT 1073741824.0f

typedef struct { float c[9]; } cvert; /* x y z w u v r g b (clip space + attributes) */

typedef struct {
    const orc_config *cfg;
    const orc_texture *textures;
    uint32_t n_textures;
    uint8_t *rgba;
    float *depth;
    uint32_t *prim;
    orc_stats st;
    int y0, y1;
} frame;

/* standard.vert:28 as the numeric contract fixes it: clip = PVM * (pos, 1), PVM = (P*V)*M,
 * each matrix-vector row summed left to right. */
static void transform_vertex(const float pvm[16], const float p[3], float out[4])
{
    for (int i = 0; i < 4; ++i)
        out[i] = ((pvm[0 + i] * p[0] + pvm[4 + i] * p[1]) + pvm[8 + i] * p[2]) + pvm[12 + i];
}

static float plane_dist(const cvert *v, int plane)
{
    float x = v->c[0], y = v->c[1], z = v->c[2], w = v->c[3];
    switch (plane) {
    case 0: return z;              /* near: 0 <= z (Vulkan depth range) */
    case 1: return w - z;          /* far */
    case 2: return GUARD * w + x;
    case 3: return GUARD * w - x;
    case 4: return GUARD * w + y;
    default: return GUARD * w - y;
    }
}

/* Intersection is always computed from the inside vertex towards the outside one so that two
 * triangles sharing an edge generate bit-identical new vertices. */
static void intersect(const cvert *in, const cvert *out, float din, float dout, cvert *r)
{
    float t = din / (din - dout);
    for (int k = 0; k < 9; ++k) r->c[k] = in->c[k] + t * (out->c[k] - in->c[k]);
}

/* Sutherland-Hodgman against near, far and the four guard-band planes, in that order. */
static int clip_polygon(cvert *poly /* capacity 12 */, int n)
{
    cvert tmp[12];
    for (int plane = 0; plane < 6; ++plane) {
        int m = 0;
        for (int i = 0; i < n; ++i) {
            const cvert *a = &poly[i], *b = &poly[(i + 1) % n];
            float da = plane_dist(a, plane), db = plane_dist(b, plane);
            if (da >= 0.0f) {
                tmp[m++] = *a;
                if (!(db >= 0.0f)) intersect(a, b, da, db, &tmp[m++]);
            } else if (db >= 0.0f) {
                intersect(b, a, db, da, &tmp[m++]);
            }
        }
        n = m;
        memcpy(poly, tmp, (size_t)n * sizeof(cvert));
        if (n < 3) return 0;
    }
    return n;
}

static void sample_texture(const orc_texture *t, float u, float v, float out[4])
{
    float fw = (float)t->width, fh = (float)t->height;
    float x = u * fw - 0.5f, y = v * fh - 0.5f;
    if (!(x >= -1.0f)) x = -1.0f;
    if (x > fw) x = fw;
    if (!(y >= -1.0f)) y = -1.0f;
    if (y > fh) y = fh;
    float fx0 = floorf(x), fy0 = floorf(y);
    float ax = x - fx0, ay = y - fy0;
    int ix0 = (int)fx0, iy0 = (int)fy0, ix1 = ix0 + 1, iy1 = iy0 + 1;
    int mw = (int)t->width - 1, mh = (int)t->height - 1;
    if (ix0 < 0) ix0 = 0; if (ix0 > mw) ix0 = mw;
    if (ix1 < 0) ix1 = 0; if (ix1 > mw) ix1 = mw;
    if (iy0 < 0) iy0 = 0; if (iy0 > mh) iy0 = mh;
    if (iy1 < 0) iy1 = 0; if (iy1 > mh) iy1 = mh;
    const uint8_t *t00 = t->texels + 4 * ((size_t)iy0 * t->width + ix0);
    const uint8_t *t10 = t->texels + 4 * ((size_t)iy0 * t->width + ix1);
    const uint8_t *t01 = t->texels + 4 * ((size_t)iy1 * t->width + ix0);
    const uint8_t *t11 = t->texels + 4 * ((size_t)iy1 * t->width + ix1);
    for (int k = 0; k < 4; ++k) {
        const float *lut = (k < 3) ? ORC_SRGB_DECODE : ORC_UNORM8; /* sRGB decode before filtering */
        float c00 = lut[t00[k]], c10 = lut[t10[k]], c01 = lut[t01[k]], c11 = lut[t11[k]];
        float top = c00 + ax * (c10 - c00);
        float bot = c01 + ax * (c11 - c01);
        out[k] = top + ay * (bot - top);
    }
}

/* ---- the lighting extension (no reference counterpart; definition at sxe_set_light in include/sxe.h) ---- */
typedef struct { float l[3], v[3], h[3]; int ok; } light_vectors;

static int unit_vector(const float v[3], float out[3])
{
    float len = sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
    if (!(len > 0.0f) || !(len <= 3.0e38f)) return 0;
    out[0] = v[0] / len; out[1] = v[1] / len; out[2] = v[2] / len;
    return 1;
}

static light_vectors make_light_vectors(const orc_config *cfg)
{
    light_vectors L;
    memset(&L, 0, sizeof L);
    L.ok = unit_vector(cfg->to_light, L.l) && unit_vector(cfg->to_viewer, L.v) && cfg->shininess_log2 <= 16;
    if (L.ok) {
        float sum[3] = { L.l[0] + L.v[0], L.l[1] + L.v[1], L.l[2] + L.v[2] };
        if (!unit_vector(sum, L.h)) L.h[0] = L.h[1] = L.h[2] = 0.0f;
    }
    return L;
}

/* The two terms of one triangle: rgb = texture.rgb * out[0] + out[1]. */
static void light_terms(const orc_config *cfg, const light_vectors *L, const float m[16], const float *p0,
                        const float *p1, const float *p2, float out[2])
{
    const float *p[3] = { p0, p1, p2 };
    float w[3][3];
    for (int k = 0; k < 3; ++k)
        for (int r = 0; r < 3; ++r) w[k][r] = ((m[r] * p[k][0] + m[4 + r] * p[k][1]) + m[8 + r] * p[k][2]) + m[12 + r];
    float e1x = w[1][0] - w[0][0], e1y = w[1][1] - w[0][1], e1z = w[1][2] - w[0][2];
    float e2x = w[2][0] - w[0][0], e2y = w[2][1] - w[0][1], e2z = w[2][2] - w[0][2];
    float nx = e1y * e2z - e1z * e2y, ny = e1z * e2x - e1x * e2z, nz = e1x * e2y - e1y * e2x;
    float len = sqrtf((nx * nx + ny * ny) + nz * nz);
    out[0] = cfg->ambient; out[1] = 0.0f;
    if (!(len > 0.0f) || !(len <= 3.0e38f)) return;
    nx = nx / len; ny = ny / len; nz = nz / len;
    float facing = (nx * L->v[0] + ny * L->v[1]) + nz * L->v[2];
    if (facing < 0.0f) { nx = -nx; ny = -ny; nz = -nz; }
    float ndl = (nx * L->l[0] + ny * L->l[1]) + nz * L->l[2];
    float ndh = (nx * L->h[0] + ny * L->h[1]) + nz * L->h[2];
    if (!(ndl > 0.0f)) ndl = 0.0f;
    if (!(ndh > 0.0f)) ndh = 0.0f;
    float spec = ndh;
    for (uint32_t k = 0; k < cfg->shininess_log2; ++k) spec = spec * spec;
    out[0] = cfg->ambient + cfg->diffuse * ndl;
    out[1] = cfg->specular * spec;
}

/* One (sub-)triangle with clip-space vertices: setup, coverage, depth test, shade, blend. */
static void raster_triangle(frame *f, const cvert tri[3], uint32_t prim_id, int32_t texture, const float lit[2])
{
    const orc_config *cfg = f->cfg;
    float half_w = (float)cfg->width * 0.5f, half_h = (float)cfg->height * 0.5f;
    int32_t X[3], Y[3];
    float zn[3], rw[3], att[3][5]; /* attributes premultiplied by 1/w: u v r g b */
    f->st.subtris++;
    for (int k = 0; k < 3; ++k) {
        float w = tri[k].c[3];
        if (!(w > 0.0f)) { f->st.subtris_culled++; return; }
        rw[k] = 1.0f / w;
        float xn = tri[k].c[0] * rw[k], yn = tri[k].c[1] * rw[k];
        zn[k] = tri[k].c[2] * rw[k];
        float xs = ((xn + 1.0f) * half_w) * (float)SUBPIX;
        float ys = ((yn + 1.0f) * half_h) * (float)SUBPIX;
        if (!(fabsf(xs) <= SNAP_LIMIT) || !(fabsf(ys) <= SNAP_LIMIT)) { f->st.subtris_culled++; return; }
        X[k] = (int32_t)rintf(xs);
        Y[k] = (int32_t)rintf(ys);
        for (int a = 0; a < 5; ++a) att[k][a] = tri[k].c[4 + a] * rw[k];
    }
    int64_t area2 = (int64_t)(X[1] - X[0]) * (int64_t)(Y[2] - Y[0]) - (int64_t)(X[2] - X[0]) * (int64_t)(Y[1] - Y[0]);
    if (area2 == 0) { f->st.subtris_culled++; return; }
    int i1 = 1, i2 = 2;
    if (area2 < 0) { i1 = 2; i2 = 1; area2 = -area2; } /* no face culling: reorient */
    int32_t X0 = X[0], Y0 = Y[0], X1 = X[i1], Y1 = Y[i1], X2 = X[i2], Y2 = Y[i2];
    int32_t xmin = X0 < X1 ? X0 : X1; if (X2 < xmin) xmin = X2;
    int32_t xmax = X0 > X1 ? X0 : X1; if (X2 > xmax) xmax = X2;
    int32_t ymin = Y0 < Y1 ? Y0 : Y1; if (Y2 < ymin) ymin = Y2;
    int32_t ymax = Y0 > Y1 ? Y0 : Y1; if (Y2 > ymax) ymax = Y2;
    /* pixel (i,j) is sampled at (256 i + 128, 256 j + 128) */
    int32_t px0 = (xmin + 127) >> 8, px1 = (xmax - 128) >> 8;
    int32_t py0 = (ymin + 127) >> 8, py1 = (ymax - 128) >> 8;
    if (px0 < 0) px0 = 0;
    if (px1 > (int32_t)cfg->width - 1) px1 = (int32_t)cfg->width - 1;
    if (py0 < f->y0) py0 = f->y0;
    if (py1 > f->y1 - 1) py1 = f->y1 - 1;
    if (px0 > px1 || py0 > py1) { f->st.subtris_culled++; return; }

    float rA = 1.0f / (float)area2;
    float z0 = zn[0], dz1 = zn[i1] - zn[0], dz2 = zn[i2] - zn[0];
    float iw0 = rw[0], diw1 = rw[i1] - rw[0], diw2 = rw[i2] - rw[0];
    float a0[5], da1[5], da2[5];
    for (int a = 0; a < 5; ++a) {
        a0[a] = att[0][a];
        da1[a] = att[i1][a] - att[0][a];
        da2[a] = att[i2][a] - att[0][a];
    }
    /* edges: e0 = v1->v2, e1 = v2->v0, e2 = v0->v1; interior has every edge function > 0 */
    int32_t ex[3] = { X2 - X1, X0 - X2, X1 - X0 };
    int32_t ey[3] = { Y2 - Y1, Y0 - Y2, Y1 - Y0 };
    int32_t ox[3] = { X1, X2, X0 }, oy[3] = { Y1, Y2, Y0 };
    int tl[3];
    for (int e = 0; e < 3; ++e) tl[e] = (ey[e] < 0) || (ey[e] == 0 && ex[e] > 0); /* top-left rule */

    uint64_t covered_before = f->st.fragments_covered;
    for (int32_t j = py0; j <= py1; ++j) {
        for (int32_t i = px0; i <= px1; ++i) {
            int64_t sx = 256 * (int64_t)i + 128, sy = 256 * (int64_t)j + 128;
            int64_t E[3];
            int inside = 1;
            f->st.fragments_tested++;
            for (int e = 0; e < 3; ++e) {
                E[e] = (int64_t)ex[e] * (sy - oy[e]) - (int64_t)ey[e] * (sx - ox[e]);
                if (!(E[e] > 0 || (E[e] == 0 && tl[e]))) inside = 0;
            }
            if (!inside) continue;
            f->st.fragments_covered++;
            float l1 = (float)E[1] * rA, l2 = (float)E[2] * rA;
            float z = (z0 + l1 * dz1) + l2 * dz2;
            if (!(z > 0.0f)) z = 0.0f;
            if (!(z < 1.0f)) continue; /* can never be Less than the cleared 1.0 */
            size_t p = (size_t)j * cfg->width + (size_t)i;
            if (!(z < f->depth[p])) continue; /* Comparison::Less, src/renderer/drawer.rs:700 */
            f->st.fragments_passed++;
            f->depth[p] = z;
            f->prim[p] = prim_id;

            float iw = (iw0 + l1 * diw1) + l2 * diw2;
            float riw = 1.0f / iw;
            float at[5];
            for (int a = 0; a < 5; ++a) at[a] = ((a0[a] + l1 * da1[a]) + l2 * da2[a]) * riw;
            float src[4];
            if (cfg->shading == ORC_SHADE_VERTEX_COLOR) { /* standard.frag:13 (commented alternative) */
                src[0] = at[2]; src[1] = at[3]; src[2] = at[4]; src[3] = 1.0f;
            } else {
                sample_texture(&f->textures[texture], at[0], at[1], src); /* standard.frag:12 */
                if (cfg->shading == ORC_SHADE_COLOR_TIMES_TEXTURE) { /* ui.frag:12 */
                    src[0] = at[2] * src[0]; src[1] = at[3] * src[1]; src[2] = at[4] * src[2];
                } else if (cfg->shading == ORC_SHADE_LIT) { /* (extension) */
                    src[0] = src[0] * lit[0] + lit[1]; src[1] = src[1] * lit[0] + lit[1]; src[2] = src[2] * lit[0] + lit[1];
                }
            }
            /* BlendState::ALPHA on an sRGB8 target: the destination is what the target holds. */
            uint8_t *px = f->rgba + 4 * p;
            float sa = src[3], ia = 1.0f - sa;
            float dr = ORC_SRGB_DECODE[px[0]], dg = ORC_SRGB_DECODE[px[1]], db = ORC_SRGB_DECODE[px[2]];
            float da = ORC_UNORM8[px[3]];
            px[0] = orc_srgb_encode(src[0] * sa + dr * ia);
            px[1] = orc_srgb_encode(src[1] * sa + dg * ia);
            px[2] = orc_srgb_encode(src[2] * sa + db * ia);
            px[3] = unorm8_encode(sa + da * ia);
        }
    }
    if (f->st.fragments_covered == covered_before) f->st.subtris_empty++;
}

static int validate(const orc_config *cfg, const orc_mesh *meshes, uint32_t n_meshes,
                    const orc_object *objects, uint32_t n_objects, const uint32_t *indices,
                    const orc_texture *textures, uint32_t n_textures)
{
    if (!cfg || cfg->width == 0 || cfg->height == 0 || cfg->width > 8192 || cfg->height > 8192) return -1;
    if (cfg->scissor_y0 > cfg->scissor_y1 || cfg->scissor_y1 > cfg->height) return -1;
    if (cfg->shading > ORC_SHADE_LIT) return -1;
    if (cfg->shading == ORC_SHADE_LIT && !make_light_vectors(cfg).ok) return -1;
    for (uint32_t m = 0; m < n_meshes; ++m) {
        if (meshes[m].n_indices % 3) return -1;
        for (uint32_t k = 0; k < meshes[m].n_indices; ++k)
            if (indices[meshes[m].first_index + k] >= meshes[m].n_vertices) return -1;
    }
    for (uint32_t o = 0; o < n_objects; ++o) {
        if (objects[o].mesh >= n_meshes) return -1;
        if (cfg->shading != ORC_SHADE_VERTEX_COLOR && (objects[o].rendered || (cfg->flags & ORC_FLAG_REFERENCE_INDEXING)) &&
            (objects[o].texture < 0 || (uint32_t)objects[o].texture >= n_textures)) return -1;
    }
    for (uint32_t t = 0; t < n_textures; ++t)
        if (!textures[t].texels || textures[t].width == 0 || textures[t].height == 0) return -1;
    return 0;
}

/* One recorded draw_indexed (src/renderer/drawer.rs:345): a range of the index buffer, which holds
 * the drawables' indices one after the other in caller order, rebased (src/renderer/drawer.rs:181-193),
 * the model matrix bound at that moment (:331-341) and the group's texture (:319-324). */
typedef struct {
    uint64_t first_index;
    uint32_t n_indices;
    uint32_t matrix;
    int32_t texture;
} draw_cmd;

/* The draws of update_drawables + generate_cmd_buffers (src/renderer/drawer.rs:481-487,313-347):
 * texture groups ascending, drawable order inside a group, unrendered meshes skipped.
 * cat_off[o] is where drawable o's indices start in the index buffer.
 * Default: every object draws its own index range under its own model matrix -- what the
 * reference means.  ORC_FLAG_REFERENCE_INDEXING: what it records: the matrix index is the mesh's
 * position inside its texture group (`i` of `enumerate`, :326,331, unrendered meshes counted), the
 * index range starts at a cursor that only advances by the draws recorded so far, in group order
 * (:313,345-346), and an unrendered mesh leaves the cursor where it is (:327-329). */
static draw_cmd *draw_list(const orc_config *cfg, const orc_mesh *meshes, const orc_object *objects, uint32_t n,
                           const uint64_t *cat_off, uint32_t *n_draws)
{
    draw_cmd *draws = (draw_cmd *)malloc((size_t)(n ? n : 1) * sizeof(draw_cmd));
    uint32_t m = 0, seen = 0;
    uint64_t cursor = 0; /* current_mesh_index, :313 */
    int compat = (cfg->flags & ORC_FLAG_REFERENCE_INDEXING) != 0;
    int32_t cur = 0x7fffffff;
    for (uint32_t o = 0; o < n; ++o)
        if (objects[o].texture < cur) cur = objects[o].texture;
    /* stable: one pass over the objects per distinct texture id, ascending */
    while (seen < n) {
        int32_t next = 0x7fffffff;
        uint32_t i = 0; /* position inside the group */
        for (uint32_t o = 0; o < n; ++o) {
            if (objects[o].texture == cur) {
                ++seen;
                if (objects[o].rendered) {
                    uint32_t n_idx = meshes[objects[o].mesh].n_indices;
                    draws[m].n_indices = n_idx;
                    draws[m].texture = cur;
                    draws[m].matrix = compat ? i : o;
                    draws[m].first_index = compat ? cursor : cat_off[o];
                    cursor += n_idx;
                    ++m;
                }
                ++i;
            } else if (objects[o].texture > cur && objects[o].texture < next) {
                next = objects[o].texture;
            }
        }
        if (next == 0x7fffffff) break;
        cur = next;
    }
    *n_draws = m;
    return draws;
}

static int render_band(const orc_config *cfg, const orc_vertex *vertices, const uint32_t *indices,
                       const orc_mesh *meshes, const orc_object *objects, uint32_t n_objects,
                       const float *models, const float view[16], const float proj[16],
                       const orc_texture *textures, uint32_t n_textures, uint8_t *out_rgba,
                       float *out_depth, uint32_t *out_prim, float *out_clip, int y0, int y1,
                       orc_stats *stats)
{
    frame f;
    memset(&f, 0, sizeof f);
    f.cfg = cfg; f.textures = textures; f.n_textures = n_textures;
    f.rgba = out_rgba; f.depth = out_depth; f.prim = out_prim; f.y0 = y0; f.y1 = y1;

    /* clear: colour attachment holds the encoded clear colour, depth 1.0 (drawer.rs:307-308) */
    uint8_t cc[4] = { orc_srgb_encode(cfg->clear[0]), orc_srgb_encode(cfg->clear[1]),
                      orc_srgb_encode(cfg->clear[2]), unorm8_encode(cfg->clear[3]) };
    for (int j = y0; j < y1; ++j)
        for (uint32_t i = 0; i < cfg->width; ++i) {
            size_t p = (size_t)j * cfg->width + i;
            memcpy(out_rgba + 4 * p, cc, 4);
            out_depth[p] = 1.0f;
            out_prim[p] = 0xFFFFFFFFu;
        }

    float pv[16];
    orc_mat4_mul(proj, view, pv);

    /* clip positions are reported in the layout of the reference's concatenated vertex buffer:
     * drawables in caller order (src/renderer/drawer.rs:173-179) */
    if (out_clip) {
        size_t base = 0;
        for (uint32_t o = 0; o < n_objects; ++o) {
            const orc_mesh *m = &meshes[objects[o].mesh];
            float pvm[16];
            orc_mat4_mul(pv, models + 16 * (size_t)o, pvm);
            for (uint32_t v = 0; v < m->n_vertices; ++v)
                transform_vertex(pvm, vertices[m->first_vertex + v].pos, out_clip + 4 * (base + v));
            base += m->n_vertices;
        }
    }

    light_vectors light = make_light_vectors(cfg);
    uint64_t *cat_off = (uint64_t *)malloc(((size_t)n_objects + 1) * sizeof(uint64_t));
    cat_off[0] = 0;
    for (uint32_t o = 0; o < n_objects; ++o) cat_off[o + 1] = cat_off[o] + meshes[objects[o].mesh].n_indices;
    uint32_t n_draws = 0;
    draw_cmd *draws = draw_list(cfg, meshes, objects, n_objects, cat_off, &n_draws);
    uint32_t prim_id = 0;
    for (uint32_t d = 0; d < n_draws; ++d) {
        const float *model = models + 16 * (size_t)draws[d].matrix;
        float pvm[16];
        orc_mat4_mul(pv, model, pvm);
        /* the drawable whose indices lie at the start of the range (binary search; then forwards) */
        uint32_t k = 0;
        {
            uint32_t lo = 0, hi = n_objects;
            while (hi - lo > 1) {
                uint32_t mid = lo + (hi - lo) / 2;
                if (cat_off[mid] <= draws[d].first_index) lo = mid; else hi = mid;
            }
            k = lo;
        }
        for (uint32_t t = 0; t < draws[d].n_indices / 3; ++t, ++prim_id) {
            cvert poly[12];
            float lit[2] = { 0.0f, 0.0f };
            const orc_vertex *corner[3];
            /* index fetch: position p of the index buffer belongs to drawable k, whose vertices the
             * rebased index value selects from the vertex buffer (src/renderer/drawer.rs:173-193) */
            uint64_t p = draws[d].first_index + 3 * (uint64_t)t;
            while (p >= cat_off[k + 1]) ++k;
            const orc_mesh *m = &meshes[objects[k].mesh];
            for (int c = 0; c < 3; ++c)
                corner[c] = &vertices[m->first_vertex + indices[m->first_index + (uint32_t)(p - cat_off[k]) + (uint32_t)c]];
            f.st.prims++;
            if (cfg->shading == ORC_SHADE_LIT)
                light_terms(cfg, &light, model, corner[0]->pos, corner[1]->pos, corner[2]->pos, lit);
            for (int c = 0; c < 3; ++c) {
                const orc_vertex *v = corner[c];
                transform_vertex(pvm, v->pos, poly[c].c);
                poly[c].c[4] = v->uv[0]; poly[c].c[5] = v->uv[1];
                poly[c].c[6] = v->color[0]; poly[c].c[7] = v->color[1]; poly[c].c[8] = v->color[2];
            }
            if (!(cfg->flags & ORC_FLAG_NO_TRIVIAL_REJECT)) {
                /* all three vertices outside one plane of the true view volume */
                int rej = 0;
                for (int pl = 0; pl < 6 && !rej; ++pl) {
                    int out = 0;
                    for (int k = 0; k < 3; ++k) {
                        float x = poly[k].c[0], y = poly[k].c[1], z = poly[k].c[2], w = poly[k].c[3];
                        float dd = pl == 0 ? z : pl == 1 ? w - z : pl == 2 ? w + x : pl == 3 ? w - x : pl == 4 ? w + y : w - y;
                        out += !(dd >= 0.0f);
                    }
                    rej = (out == 3);
                }
                if (rej) { f.st.prims_rejected++; continue; }
            }
            int need_clip = 0;
            for (int k = 0; k < 3; ++k)
                for (int pl = 0; pl < 6; ++pl)
                    if (!(plane_dist(&poly[k], pl) >= 0.0f)) need_clip = 1;
            int n = 3;
            if (need_clip) {
                f.st.prims_clipped++;
                n = clip_polygon(poly, 3);
                if (n < 3) continue;
            }
            for (int s = 1; s + 1 < n; ++s) { /* fan; sub-triangle index s-1 */
                cvert tri[3] = { poly[0], poly[s], poly[s + 1] };
                raster_triangle(&f, tri, prim_id, draws[d].texture, lit);
            }
        }
    }
    free(draws);
    free(cat_off);
    /* the target's byte order: blending worked on the channels by name; a Bgra8 target holds them
     * with red and blue exchanged (src/renderer/presenter.rs:472-478) */
    if (cfg->flags & ORC_FLAG_BGRA)
        for (int j = y0; j < y1; ++j)
            for (uint32_t i = 0; i < cfg->width; ++i) {
                uint8_t *px = out_rgba + 4 * ((size_t)j * cfg->width + i), t = px[0];
                px[0] = px[2];
                px[2] = t;
            }
    if (stats) *stats = f.st;
    return 0;
}

int orc_render(const orc_config *cfg, const orc_vertex *vertices, const uint32_t *indices,
               const orc_mesh *meshes, uint32_t n_meshes, const orc_object *objects,
               uint32_t n_objects, const float *models, const float view[16],
               const float proj[16], const orc_texture *textures, uint32_t n_textures,
               uint8_t *out_rgba, float *out_depth, uint32_t *out_prim, float *out_clip,
               orc_stats *stats)
{
    if (validate(cfg, meshes, n_meshes, objects, n_objects, indices, textures, n_textures)) return -1;
    return render_band(cfg, vertices, indices, meshes, objects, n_objects, models, view, proj,
                       textures, n_textures, out_rgba, out_depth, out_prim, out_clip,
                       (int)cfg->scissor_y0, (int)cfg->scissor_y1, stats);
}

typedef struct {
    const orc_config *cfg; const orc_vertex *vertices; const uint32_t *indices;
    const orc_mesh *meshes; const orc_object *objects; uint32_t n_objects;
    const float *models; const float *view; const float *proj;
    const orc_texture *textures; uint32_t n_textures;
    uint8_t *rgba; float *depth; uint32_t *prim; int y0, y1;
} band_job;

static void *band_main(void *arg)
{
    band_job *j = (band_job *)arg;
    render_band(j->cfg, j->vertices, j->indices, j->meshes, j->objects, j->n_objects, j->models,
                j->view, j->proj, j->textures, j->n_textures, j->rgba, j->depth, j->prim, NULL,
                j->y0, j->y1, NULL);
    return NULL;
}

int orc_render_mt(const orc_config *cfg, const orc_vertex *vertices, const uint32_t *indices,
                  const orc_mesh *meshes, uint32_t n_meshes, const orc_object *objects,
                  uint32_t n_objects, const float *models, const float view[16],
                  const float proj[16], const orc_texture *textures, uint32_t n_textures,
                  uint8_t *out_rgba, float *out_depth, uint32_t *out_prim, int n_threads)
{
    if (validate(cfg, meshes, n_meshes, objects, n_objects, indices, textures, n_textures)) return -1;
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    pthread_t th[256];
    band_job jobs[256];
    int rows = (int)(cfg->scissor_y1 - cfg->scissor_y0);
    for (int t = 0; t < n_threads; ++t) {
        band_job *j = &jobs[t];
        j->cfg = cfg; j->vertices = vertices; j->indices = indices; j->meshes = meshes;
        j->objects = objects; j->n_objects = n_objects; j->models = models; j->view = view;
        j->proj = proj; j->textures = textures; j->n_textures = n_textures;
        j->rgba = out_rgba; j->depth = out_depth; j->prim = out_prim;
        j->y0 = (int)cfg->scissor_y0 + (int)((int64_t)rows * t / n_threads);
        j->y1 = (int)cfg->scissor_y0 + (int)((int64_t)rows * (t + 1) / n_threads);
        if (pthread_create(&th[t], NULL, band_main, j)) return -2;
    }
    for (int t = 0; t < n_threads; ++t) pthread_join(th[t], NULL);
    return 0;
}
