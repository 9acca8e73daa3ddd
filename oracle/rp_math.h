/* oracle/rp_math.h — small vector/quaternion helpers for the CPU oracle (TEST INFRASTRUCTURE ONLY).
 * Quaternions are xyzw as in PyBullet.  Euler conversions restate the formulas SURVEY.md App. E recalls
 * from bullet3 (b3Quaternion::setEulerZYX / pybullet.c getEulerFromQuaternion) — unverified against a
 * live PyBullet ("parity unpinned"), pinned here by analytic identities in tests/test_oracle_math.py. */
#ifndef RP_ORACLE_MATH_H
#define RP_ORACLE_MATH_H
#include <math.h>

#ifdef RP_FLOAT
typedef float real;
#define R_SQRT sqrtf
#define R_EPS 1.1920929e-7f
#define R_SIN sinf
#define R_COS cosf
#define R_ATAN2 atan2f
#define R_ASIN asinf
#define R_ACOS acosf
#define R_FABS fabsf
#define R_FMOD fmodf
#define R_FLOOR floorf
#define R_FMA fmaf
#else
typedef double real;
#define R_SQRT sqrt
#define R_EPS 2.220446049250313e-16
#define R_SIN sin
#define R_COS cos
#define R_ATAN2 atan2
#define R_ASIN asin
#define R_ACOS acos
#define R_FABS fabs
#define R_FMOD fmod
#define R_FLOOR floor
#define R_FMA fma
#endif

#define RP_PI ((real)3.14159265358979323846)
/* a . b as the HIP library's hull scans form it (rp_kernels.cuh hull_coord, the GJK support scan): z, y, x folded in by fused multiply-adds - one rounding per step,
 * the same vertex wins a near-tie on both sides */
#define R_DOT3_FMA(a0, a1, a2, b0, b1, b2) R_FMA((a2), (b2), R_FMA((a1), (b1), (a0) * (b0)))

static inline void v3set(real* o, real x, real y, real z) { o[0] = x; o[1] = y; o[2] = z; }
static inline void v3cpy(real* o, const real* a) { o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; }
static inline void v3add(real* o, const real* a, const real* b) { o[0] = a[0] + b[0]; o[1] = a[1] + b[1]; o[2] = a[2] + b[2]; }
static inline void v3sub(real* o, const real* a, const real* b) { o[0] = a[0] - b[0]; o[1] = a[1] - b[1]; o[2] = a[2] - b[2]; }
static inline void v3scale(real* o, const real* a, real s) { o[0] = a[0] * s; o[1] = a[1] * s; o[2] = a[2] * s; }
static inline void v3axpy(real* o, real s, const real* a) { o[0] += s * a[0]; o[1] += s * a[1]; o[2] += s * a[2]; }
/* products of 3-vectors and 3 x 3 matrices accumulate through fused multiply-adds in the HIP library's order (rp_math.cuh dot / cross / mulv / tmulv / mul: x first,
 * then y, then z folded in): one rounding convention on both sides (round 5) */
static inline real v3dot(const real* a, const real* b) { return R_FMA(a[2], b[2], R_FMA(a[1], b[1], a[0] * b[0])); }
static inline real v3norm(const real* a) { return R_SQRT(v3dot(a, a)); }
static inline void v3cross(real* o, const real* a, const real* b) {
  real x = R_FMA(a[1], b[2], -(a[2] * b[1])), y = R_FMA(a[2], b[0], -(a[0] * b[2])), z = R_FMA(a[0], b[1], -(a[1] * b[0]));
  o[0] = x; o[1] = y; o[2] = z;
}
/* 3x3 row-major */
static inline void m3mulv(real* o, const real* M, const real* v) {
  real x = R_FMA(M[2], v[2], R_FMA(M[1], v[1], M[0] * v[0])), y = R_FMA(M[5], v[2], R_FMA(M[4], v[1], M[3] * v[0])),
       z = R_FMA(M[8], v[2], R_FMA(M[7], v[1], M[6] * v[0]));
  o[0] = x; o[1] = y; o[2] = z;
}
static inline void m3tmulv(real* o, const real* M, const real* v) {
  real x = R_FMA(M[6], v[2], R_FMA(M[3], v[1], M[0] * v[0])), y = R_FMA(M[7], v[2], R_FMA(M[4], v[1], M[1] * v[0])),
       z = R_FMA(M[8], v[2], R_FMA(M[5], v[1], M[2] * v[0]));
  o[0] = x; o[1] = y; o[2] = z;
}
static inline void m3mul(real* o, const real* A, const real* B) {
  real t[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) t[3 * i + j] = R_FMA(A[3 * i + 2], B[6 + j], R_FMA(A[3 * i + 1], B[3 + j], A[3 * i] * B[j]));
  for (int i = 0; i < 9; i++) o[i] = t[i];
}
static inline void m3ident(real* M) { for (int i = 0; i < 9; i++) M[i] = (i % 4 == 0) ? (real)1 : (real)0; }
/* rotation about unit axis a by angle q (Rodrigues) */
static inline void m3axis_angle(real* M, const real* a, real q) {
  real c = R_COS(q), s = R_SIN(q), t = 1 - c;
  M[0] = t * a[0] * a[0] + c;        M[1] = t * a[0] * a[1] - s * a[2]; M[2] = t * a[0] * a[2] + s * a[1];
  M[3] = t * a[0] * a[1] + s * a[2]; M[4] = t * a[1] * a[1] + c;        M[5] = t * a[1] * a[2] - s * a[0];
  M[6] = t * a[0] * a[2] - s * a[1]; M[7] = t * a[1] * a[2] + s * a[0]; M[8] = t * a[2] * a[2] + c;
}
static inline void quat_to_m3(real* M, const real* q) {
  real x = q[0], y = q[1], z = q[2], w = q[3];
  real d = x * x + y * y + z * z + w * w, s = (real)2 / d;   /* btMatrix3x3::setRotation: tolerant of non-unit q */
  real xs = x * s, ys = y * s, zs = z * s, wx = w * xs, wy = w * ys, wz = w * zs, xx = x * xs, xy = x * ys, xz = x * zs,
       yy = y * ys, yz = y * zs, zz = z * zs;
  M[0] = 1 - (yy + zz); M[1] = xy - wz;       M[2] = xz + wy;
  M[3] = xy + wz;       M[4] = 1 - (xx + zz); M[5] = yz - wx;
  M[6] = xz - wy;       M[7] = yz + wx;       M[8] = 1 - (xx + yy);
}
/* btMatrix3x3::getRotation */
static inline void m3_to_quat(real* q, const real* M) {
  real tr = M[0] + M[4] + M[8];
  if (tr > 0) {
    real s = R_SQRT(tr + 1);
    q[3] = s * (real)0.5; s = (real)0.5 / s;
    q[0] = (M[7] - M[5]) * s; q[1] = (M[2] - M[6]) * s; q[2] = (M[3] - M[1]) * s;
  } else {
    int i = M[0] < M[4] ? (M[4] < M[8] ? 2 : 1) : (M[0] < M[8] ? 2 : 0);
    int j = (i + 1) % 3, k = (i + 2) % 3;
    real s = R_SQRT(M[4 * i] - M[4 * j] - M[4 * k] + 1);
    q[i] = s * (real)0.5; s = (real)0.5 / s;
    q[3] = (M[3 * k + j] - M[3 * j + k]) * s;
    q[j] = (M[3 * j + i] + M[3 * i + j]) * s;
    q[k] = (M[3 * k + i] + M[3 * i + k]) * s;
  }
}
static inline void quat_mul(real* o, const real* a, const real* b) { /* a (x) b, xyzw */
  real x = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  real y = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
  real z = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
  real w = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  o[0] = x; o[1] = y; o[2] = z; o[3] = w;
}
/* pybullet getQuaternionFromEuler */
static inline void quat_from_euler(real* q, const real* rpy) {
  real hr = rpy[0] * (real)0.5, hp = rpy[1] * (real)0.5, hy = rpy[2] * (real)0.5;
  real cr = R_COS(hr), sr = R_SIN(hr), cp = R_COS(hp), sp = R_SIN(hp), cy = R_COS(hy), sy = R_SIN(hy);
  q[0] = sr * cp * cy - cr * sp * sy;
  q[1] = cr * sp * cy + sr * cp * sy;
  q[2] = cr * cp * sy - sr * sp * cy;
  q[3] = cr * cp * cy + sr * sp * sy;
}
/* pybullet getEulerFromQuaternion — no normalisation of q (quirk F2 relies on it) */
static inline void euler_from_quat(real* rpy, const real* q) {
  real x = q[0], y = q[1], z = q[2], w = q[3];
  real sarg = (real)-2 * (x * z - w * y);
  if (sarg <= (real)-0.99999) {
    rpy[0] = 0; rpy[1] = (real)-0.5 * RP_PI; rpy[2] = 2 * R_ATAN2(x, -y);
  } else if (sarg >= (real)0.99999) {
    rpy[0] = 0; rpy[1] = (real)0.5 * RP_PI; rpy[2] = 2 * R_ATAN2(-x, y);
  } else {
    real sqx = x * x, sqy = y * y, sqz = z * z, sqw = w * w;
    rpy[0] = R_ATAN2(2 * (y * z + w * x), sqw - sqx - sqy + sqz);
    rpy[1] = R_ASIN(sarg);
    rpy[2] = R_ATAN2(2 * (x * y + w * z), sqw + sqx - sqy - sqz);
  }
}
#endif
