/* rp_math.cuh — fp32 device math for the playroom kernels (gfx950).  Quaternions xyzw (PyBullet order). */
#pragma once
#include <hip/hip_runtime.h>

#define RP_PI_F 3.14159265358979323846f

struct V3 { float x, y, z; };
struct M3 { float m[9]; };      /* row-major */
struct Q4 { float x, y, z, w; };

__device__ __forceinline__ V3 mk3(float x, float y, float z) { V3 r = {x, y, z}; return r; }
__device__ __forceinline__ V3 ld3(const float* p) { return mk3(p[0], p[1], p[2]); }
__device__ __forceinline__ void st3(float* p, V3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ V3 operator-(V3 a) { return mk3(-a.x, -a.y, -a.z); }
/* Products of 3-vectors and 3 x 3 matrices accumulate through FUSED multiply-adds in a fixed order (x first, then y, then z folded in): the library is compiled with
 * -ffp-contract=off, so these are the only fusions there are, and oracle/rp_math.h has the same ones in v3dot / v3cross / m3mulv / m3tmulv / m3mul (round 5: the
 * rounding convention is shared, not left to either compiler). */
__device__ __forceinline__ float dot(V3 a, V3 b) { return __fmaf_rn(a.z, b.z, __fmaf_rn(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return mk3(__fmaf_rn(a.y, b.z, -(a.z * b.y)), __fmaf_rn(a.z, b.x, -(a.x * b.z)), __fmaf_rn(a.x, b.y, -(a.y * b.x))); }
__device__ __forceinline__ float norm(V3 a) { return sqrtf(dot(a, a)); }
__device__ __forceinline__ float comp(V3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

__device__ __forceinline__ M3 ldm3(const float* p) { M3 r; for (int i = 0; i < 9; i++) r.m[i] = p[i]; return r; }
__device__ __forceinline__ void stm3(float* p, const M3& a) { for (int i = 0; i < 9; i++) p[i] = a.m[i]; }
__device__ __forceinline__ M3 ident3() { M3 r = {{1, 0, 0, 0, 1, 0, 0, 0, 1}}; return r; }
__device__ __forceinline__ V3 mulv(const M3& a, V3 v) {
  return mk3(__fmaf_rn(a.m[2], v.z, __fmaf_rn(a.m[1], v.y, a.m[0] * v.x)), __fmaf_rn(a.m[5], v.z, __fmaf_rn(a.m[4], v.y, a.m[3] * v.x)),
             __fmaf_rn(a.m[8], v.z, __fmaf_rn(a.m[7], v.y, a.m[6] * v.x)));
}
__device__ __forceinline__ V3 tmulv(const M3& a, V3 v) {
  return mk3(__fmaf_rn(a.m[6], v.z, __fmaf_rn(a.m[3], v.y, a.m[0] * v.x)), __fmaf_rn(a.m[7], v.z, __fmaf_rn(a.m[4], v.y, a.m[1] * v.x)),
             __fmaf_rn(a.m[8], v.z, __fmaf_rn(a.m[5], v.y, a.m[2] * v.x)));
}
__device__ __forceinline__ M3 mul(const M3& a, const M3& b) {
  M3 r;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) r.m[3 * i + j] = __fmaf_rn(a.m[3 * i + 2], b.m[6 + j], __fmaf_rn(a.m[3 * i + 1], b.m[3 + j], a.m[3 * i] * b.m[j]));
  return r;
}
__device__ __forceinline__ V3 col(const M3& a, int i) { return mk3(a.m[i], a.m[3 + i], a.m[6 + i]); }
__device__ __forceinline__ M3 axis_angle(V3 a, float q) {
  float s, c;
  sincosf(q, &s, &c);
  float t = 1.f - c;
  M3 r = {{t * a.x * a.x + c, t * a.x * a.y - s * a.z, t * a.x * a.z + s * a.y, t * a.x * a.y + s * a.z, t * a.y * a.y + c,
           t * a.y * a.z - s * a.x, t * a.x * a.z - s * a.y, t * a.y * a.z + s * a.x, t * a.z * a.z + c}};
  return r;
}
__device__ __forceinline__ M3 quat_to_m3(Q4 q) { /* btMatrix3x3::setRotation (tolerates non-unit q) */
  float d = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w, s = 2.f / d;
  float xs = q.x * s, ys = q.y * s, zs = q.z * s, wx = q.w * xs, wy = q.w * ys, wz = q.w * zs, xx = q.x * xs, xy = q.x * ys,
        xz = q.x * zs, yy = q.y * ys, yz = q.y * zs, zz = q.z * zs;
  M3 r = {{1.f - (yy + zz), xy - wz, xz + wy, xy + wz, 1.f - (xx + zz), yz - wx, xz - wy, yz + wx, 1.f - (xx + yy)}};
  return r;
}
__device__ __forceinline__ Q4 m3_to_quat(const M3& M) { /* btMatrix3x3::getRotation; the three pivot cases written out so
                                                            that every index is static (a dynamically indexed t[] lives in scratch) */
  float tr = M.m[0] + M.m[4] + M.m[8];
  Q4 q;
  if (tr > 0.f) {
    float s = sqrtf(tr + 1.f);
    q.w = s * 0.5f; s = 0.5f / s;
    q.x = (M.m[7] - M.m[5]) * s; q.y = (M.m[2] - M.m[6]) * s; q.z = (M.m[3] - M.m[1]) * s;
  } else {
    int i = M.m[0] < M.m[4] ? (M.m[4] < M.m[8] ? 2 : 1) : (M.m[0] < M.m[8] ? 2 : 0);
    if (i == 0) {          /* j = 1, k = 2 */
      float s = sqrtf(M.m[0] - M.m[4] - M.m[8] + 1.f);
      q.x = s * 0.5f; s = 0.5f / s;
      q.w = (M.m[7] - M.m[5]) * s; q.y = (M.m[3] + M.m[1]) * s; q.z = (M.m[6] + M.m[2]) * s;
    } else if (i == 1) {   /* j = 2, k = 0 */
      float s = sqrtf(M.m[4] - M.m[8] - M.m[0] + 1.f);
      q.y = s * 0.5f; s = 0.5f / s;
      q.w = (M.m[2] - M.m[6]) * s; q.z = (M.m[7] + M.m[5]) * s; q.x = (M.m[1] + M.m[3]) * s;
    } else {               /* j = 0, k = 1 */
      float s = sqrtf(M.m[8] - M.m[0] - M.m[4] + 1.f);
      q.z = s * 0.5f; s = 0.5f / s;
      q.w = (M.m[3] - M.m[1]) * s; q.x = (M.m[2] + M.m[6]) * s; q.y = (M.m[5] + M.m[7]) * s;
    }
  }
  return q;
}
__device__ __forceinline__ Q4 qmul(Q4 a, Q4 b) {
  Q4 r = {a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
          a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
  return r;
}
__device__ __forceinline__ Q4 quat_from_euler(float r, float p, float y) { /* pybullet getQuaternionFromEuler */
  float sr, cr, sp, cp, sy, cy;
  sincosf(r * 0.5f, &sr, &cr); sincosf(p * 0.5f, &sp, &cp); sincosf(y * 0.5f, &sy, &cy);
  Q4 q = {sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy};
  return q;
}
__device__ __forceinline__ V3 euler_from_quat(float x, float y, float z, float w) { /* pybullet getEulerFromQuaternion, no normalisation */
  float sarg = -2.f * (x * z - w * y);
  if (sarg <= -0.99999f) return mk3(0.f, -0.5f * RP_PI_F, 2.f * atan2f(x, -y));
  if (sarg >= 0.99999f) return mk3(0.f, 0.5f * RP_PI_F, 2.f * atan2f(-x, y));
  float sqx = x * x, sqy = y * y, sqz = z * z, sqw = w * w;
  return mk3(atan2f(2.f * (y * z + w * x), sqw - sqx - sqy + sqz), asinf(sarg), atan2f(2.f * (x * y + w * z), sqw + sqx - sqy - sqz));
}

/* 6-vectors [ang; lin] about the per-substep reference point */
struct V6 { V3 a, l; };
__device__ __forceinline__ V6 ld6(const float* p) { V6 r = {ld3(p), ld3(p + 3)}; return r; }
__device__ __forceinline__ void st6(float* p, V6 v) { st3(p, v.a); st3(p + 3, v.l); }
__device__ __forceinline__ V6 operator+(V6 a, V6 b) { V6 r = {a.a + b.a, a.l + b.l}; return r; }
__device__ __forceinline__ V6 operator*(V6 a, float s) { V6 r = {a.a * s, a.l * s}; return r; }
__device__ __forceinline__ float dot6(V6 a, V6 b) { return dot(a.a, b.a) + dot(a.l, b.l); }
__device__ __forceinline__ V6 zero6() { V6 r = {mk3(0, 0, 0), mk3(0, 0, 0)}; return r; }
__device__ __forceinline__ V6 crm(V6 v, V6 m) { V6 r = {cross(v.a, m.a), cross(v.a, m.l) + cross(v.l, m.a)}; return r; }   /* v x m   */
__device__ __forceinline__ V6 crf(V6 v, V6 f) { V6 r = {cross(v.a, f.a) + cross(v.l, f.l), cross(v.a, f.l)}; return r; }   /* v x* f  */
/* rigid-body spatial inertia (m, h = m c, Ibar sym xx yy zz xy xz yz) times motion vector -> force vector */
__device__ __forceinline__ V6 inertia_mul(const float* I10, V6 v) {
  float mass = I10[0];
  V3 h = ld3(I10 + 1);
  V3 Iw = mk3(I10[4] * v.a.x + I10[7] * v.a.y + I10[8] * v.a.z, I10[7] * v.a.x + I10[5] * v.a.y + I10[9] * v.a.z,
              I10[8] * v.a.x + I10[9] * v.a.y + I10[6] * v.a.z);
  V6 r = {Iw + cross(h, v.l), v.l * mass - cross(h, v.a)};
  return r;
}
