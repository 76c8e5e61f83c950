// sxe_math.h -- scalar maths shared by the host side and the gfx950 kernels of libsxe_hip.
//
// Everything here is the numeric contract of the path: each expression is evaluated exactly as
// written (the library is built with -ffp-contract=off, IEEE division and square root), so the
// same inputs give the same bits on the host and on the device.  Trigonometry uses fixed
// polynomial forms instead of a maths library because library results differ in the last bit
// between platforms; cgmath (which the reference calls at src/components/transform.rs:38-42 and
// src/renderer/drawer.rs:244,391-395) would use the platform libm.
//
// Matrices are column-major: m[4*col + row].
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define SXE_HD __host__ __device__ inline

namespace sxe {

struct Mat4 {
    float m[16];
};

struct Quat { // cgmath order: scalar first
    float s, x, y, z;
};

SXE_HD float f_abs(float x) { return x < 0.0f ? -x : (x == 0.0f ? 0.0f : x); }

// sin and cos of x (radians) together: Cody-Waite reduction by multiples of pi/2 (three-part
// pi/4), degree-7 / degree-6 minimax polynomials on [-pi/4, pi/4].
SXE_HD void sincos_det(float x, float &sn, float &cs)
{
    const float ax = f_abs(x);
    if (!(ax <= 1.0e6f)) {
        sn = 0.0f;
        cs = 1.0f;
        return;
    }
    int32_t oct = (int32_t)(ax * 1.27323954473516f);
    oct = (oct + 1) & ~1;
    const float fo = (float)oct;
    const float r = ((ax - fo * 0.78515625f) - fo * 2.4187564849853515625e-4f) - fo * 3.77489497744594108e-8f;
    const float r2 = r * r;
    const float sp = ((-1.9515295891e-4f * r2 + 8.3321608736e-3f) * r2 - 1.6666654611e-1f) * r2 * r + r;
    float cp = ((2.443315711809948e-5f * r2 - 1.388731625493765e-3f) * r2 + 4.166664568298827e-2f) * r2 * r2;
    cp = cp - 0.5f * r2;
    cp = cp + 1.0f;
    const int quad = (oct >> 1) & 3;
    float a = (quad & 1) ? cp : sp;
    float b = (quad & 1) ? sp : cp;
    if (quad & 2) a = -a;
    if (quad == 1 || quad == 2) b = -b;
    sn = (x < 0.0f) ? -a : a;
    cs = b;
}

SXE_HD float atan_nonneg(float x)
{
    float base = 0.0f;
    if (x > 2.414213562373095f) {
        base = 1.5707963267948966f;
        x = -(1.0f / x);
    } else if (x > 0.4142135623730950f) {
        base = 0.7853981633974483f;
        x = (x - 1.0f) / (x + 1.0f);
    }
    const float z = x * x;
    const float p = (((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * x + x;
    return base + p;
}

SXE_HD float atan2_det(float y, float x)
{
    const float kPi = 3.14159265358979323846f;
    const float kHalfPi = 1.5707963267948966f;
    if (x == 0.0f) return y < 0.0f ? -kHalfPi : (y == 0.0f ? 0.0f : kHalfPi);
    if (y == 0.0f) return x < 0.0f ? kPi : 0.0f;
    const float q = y / x;
    const float a = q < 0.0f ? -atan_nonneg(-q) : atan_nonneg(q);
    if (x < 0.0f) return y < 0.0f ? a - kPi : a + kPi;
    return a;
}

SXE_HD float asin_det(float x)
{
    const float kHalfPi = 1.5707963267948966f;
    const float a = f_abs(x);
    if (!(a <= 1.0f)) return x < 0.0f ? -kHalfPi : kHalfPi;
    if (a < 1.0e-4f) return x;
    const bool upper = a > 0.5f;
    float z, w;
    if (upper) {
        z = 0.5f * (1.0f - a);
        w = sqrtf(z);
    } else {
        w = a;
        z = w * w;
    }
    float p = ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z + 7.4953002686e-2f) * z + 1.6666752422e-1f) * z * w + w;
    if (upper) {
        p = p + p;
        p = kHalfPi - p;
    }
    return x < 0.0f ? -p : p;
}

// Matrix4 * Matrix4 as cgmath sums it: column j = ((A.c0*B[j].x + A.c1*B[j].y) + A.c2*B[j].z) + A.c3*B[j].w.
SXE_HD Mat4 mul(const Mat4 &A, const Mat4 &B)
{
    Mat4 R;
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i)
            R.m[4 * j + i] = ((A.m[i] * B.m[4 * j] + A.m[4 + i] * B.m[4 * j + 1]) + A.m[8 + i] * B.m[4 * j + 2]) + A.m[12 + i] * B.m[4 * j + 3];
    return R;
}

SXE_HD Mat4 identity()
{
    Mat4 R;
    for (int k = 0; k < 16; ++k) R.m[k] = (k % 5 == 0) ? 1.0f : 0.0f;
    return R;
}

// Quaternion::from(Euler{Deg, Deg, Deg}); src/components/transform.rs:28,38-42.
SXE_HD Quat quat_from_euler_deg(float xd, float yd, float zd)
{
    const float kDegToRad = 0.017453292519943295f;
    float sx, cx, sy, cy, sz, cz;
    sincos_det((xd * kDegToRad) * 0.5f, sx, cx);
    sincos_det((yd * kDegToRad) * 0.5f, sy, cy);
    sincos_det((zd * kDegToRad) * 0.5f, sz, cz);
    Quat q;
    q.s = -sx * sy * sz + cx * cy * cz;
    q.x = sx * cy * cz + sy * sz * cx;
    q.y = -sx * sz * cy + sy * cx * cz;
    q.z = sx * sy * cz + sz * cx * cy;
    return q;
}

// Hamilton product; src/components/transform.rs:44.
SXE_HD Quat mul(const Quat &a, const Quat &b)
{
    Quat r;
    r.s = a.s * b.s - a.x * b.x - a.y * b.y - a.z * b.z;
    r.x = a.s * b.x + a.x * b.s + a.y * b.z - a.z * b.y;
    r.y = a.s * b.y + a.y * b.s + a.z * b.x - a.x * b.z;
    r.z = a.s * b.z + a.z * b.s + a.x * b.y - a.y * b.x;
    return r;
}

// Matrix4::from(Quaternion); src/components/transform.rs:76.
SXE_HD Mat4 mat_from_quat(const Quat &q)
{
    const float x2 = q.x + q.x, y2 = q.y + q.y, z2 = q.z + q.z;
    const float xx2 = x2 * q.x, xy2 = x2 * q.y, xz2 = x2 * q.z;
    const float yy2 = y2 * q.y, yz2 = y2 * q.z, zz2 = z2 * q.z;
    const float sy2 = y2 * q.s, sz2 = z2 * q.s, sx2 = x2 * q.s;
    Mat4 R;
    R.m[0] = 1.0f - yy2 - zz2; R.m[1] = xy2 + sz2;        R.m[2] = xz2 - sy2;         R.m[3] = 0.0f;
    R.m[4] = xy2 - sz2;        R.m[5] = 1.0f - xx2 - zz2; R.m[6] = yz2 + sx2;         R.m[7] = 0.0f;
    R.m[8] = xz2 + sy2;        R.m[9] = yz2 - sx2;        R.m[10] = 1.0f - xx2 - yy2; R.m[11] = 0.0f;
    R.m[12] = 0.0f;            R.m[13] = 0.0f;            R.m[14] = 0.0f;             R.m[15] = 1.0f;
    return R;
}

// Transform::to_ubo: I.concat(T).concat(R).concat(S) as three general products;
// src/components/transform.rs:69-80.
SXE_HD Mat4 model_matrix(const float pos[3], const float scale[3], const Quat &rot)
{
    Mat4 T = identity();
    T.m[12] = pos[0]; T.m[13] = pos[1]; T.m[14] = pos[2];
    Mat4 S = identity();
    S.m[0] = scale[0]; S.m[5] = scale[1]; S.m[10] = scale[2];
    return mul(mul(mul(identity(), T), mat_from_quat(rot)), S);
}

// Euler::from(Quaternion) in radians; src/renderer/drawer.rs:244.
SXE_HD void euler_from_quat(const Quat &q, float &ex, float &ey, float &ez)
{
    const float sqw = q.s * q.s, sqx = q.x * q.x, sqy = q.y * q.y, sqz = q.z * q.z;
    const float unit = sqx + sqz + sqy + sqw;
    const float test = q.x * q.z + q.y * q.s;
    if (test > 0.499f * unit) {
        ex = 0.0f; ey = 1.5707963267948966f; ez = atan2_det(q.x, q.s) * 2.0f;
    } else if (test < -0.499f * unit) {
        ex = 0.0f; ey = -1.5707963267948966f; ez = -atan2_det(q.x, q.s) * 2.0f;
    } else {
        ex = atan2_det(2.0f * (-q.y * q.z + q.x * q.s), 1.0f - 2.0f * (sqx + sqy));
        ey = asin_det(2.0f * (q.x * q.z + q.y * q.s));
        ez = atan2_det(2.0f * (-q.x * q.y + q.z * q.s), 1.0f - 2.0f * (sqy + sqz));
    }
}

// fps_view_matrix; src/renderer/drawer.rs:390-409.
SXE_HD Mat4 fps_view(const float eye[3], float pitch, float yaw)
{
    float sp, cp, sy, cy;
    sincos_det(pitch, sp, cp);
    sincos_det(yaw, sy, cy);
    const float X[3] = {cy, 0.0f, -sy};
    const float Y[3] = {sy * sp, cp, cy * sp};
    const float Z[3] = {sy * cp, -sp, cp * cy};
    Mat4 V;
    for (int c = 0; c < 3; ++c) {
        V.m[4 * c + 0] = X[c];
        V.m[4 * c + 1] = Y[c];
        V.m[4 * c + 2] = Z[c];
        V.m[4 * c + 3] = 0.0f;
    }
    V.m[12] = ((X[0] * eye[0] + X[1] * eye[1]) + X[2] * eye[2]) * -1.0f;
    V.m[13] = ((Y[0] * eye[0] + Y[1] * eye[1]) + Y[2] * eye[2]) * -1.0f;
    V.m[14] = ((Z[0] * eye[0] + Z[1] * eye[1]) + Z[2] * eye[2]) * -1.0f;
    V.m[15] = 1.0f;
    return V;
}

// cgmath::perspective(Deg(fov), aspect, near, far) followed by proj.y.y *= -1;
// src/renderer/drawer.rs:248-255.  OpenGL z range is kept on purpose.
SXE_HD Mat4 perspective_yflip(float fov_deg, float aspect, float zn, float zf)
{
    float s, c;
    sincos_det((fov_deg * 0.017453292519943295f) / 2.0f, s, c);
    const float f = c / s;
    Mat4 P;
    for (int k = 0; k < 16; ++k) P.m[k] = 0.0f;
    P.m[0] = f / aspect;
    P.m[5] = f * -1.0f;
    P.m[10] = (zf + zn) / (zn - zf);
    P.m[11] = -1.0f;
    P.m[14] = (2.0f * zf * zn) / (zn - zf);
    return P;
}

// Counter-based stand-in for thread_rng().gen::<f32>() (src/systems/rotation.rs:33):
// SplitMix64 finaliser over (seed, frame, index); top 24 bits scaled to [0,1).
SXE_HD float rand01(uint64_t seed, uint32_t frame, uint32_t index)
{
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (((uint64_t)frame << 32 | (uint64_t)index) + 1ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (float)(uint32_t)(z >> 40) * 5.9604644775390625e-8f;
}

} // namespace sxe
