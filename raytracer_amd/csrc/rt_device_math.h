/*
 * rt_device_math.h - double-precision 3-vector helpers for the gfx950 kernels,
 * each with the operation order of the reference's tuple functions
 * (RayTracer/RayTracer.c:304-365, 419-424, 1065-1069).
 *
 * The reference carries a 4th lane w through every tuple operation.  On this
 * path w is exactly 0 for vectors and exactly 1 for points (the host rejects
 * non-affine transforms), so the w products are exact zeros and adding them
 * cannot change a sum; they are left out here.
 *
 * Compile with -ffp-contract=off: a*b+c must round twice, as it does in the
 * reference's x86-64 SSE2 code.
 */
#ifndef RT_DEVICE_MATH_H
#define RT_DEVICE_MATH_H

#include <hip/hip_runtime.h>

struct d3 {
  double x, y, z;
};

__device__ __forceinline__ d3 mk3(double x, double y, double z) {
  d3 r;
  r.x = x; r.y = y; r.z = z;
  return r;
}
__device__ __forceinline__ d3 add3(d3 a, d3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }   /* :304 */
__device__ __forceinline__ d3 sub3(d3 a, d3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }   /* :309 */
__device__ __forceinline__ d3 neg3(d3 a) { return mk3(0.0 - a.x, 0.0 - a.y, 0.0 - a.z); }          /* :314 */
__device__ __forceinline__ d3 scale3(d3 a, double s) { return mk3(a.x * s, a.y * s, a.z * s); }    /* :320 */
__device__ __forceinline__ d3 mul3(d3 a, d3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }    /* :330 */
__device__ __forceinline__ double dot3(d3 a, d3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; } /* :350 */
__device__ __forceinline__ d3 cross3(d3 a, d3 b) {                                                 /* :359 */
  return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ double mag3(d3 a) { return sqrt((a.x * a.x + a.y * a.y) + a.z * a.z); } /* :335 */

/* Three correctly rounded quotients by one divisor.  a / m in IEEE double is what the compiler's own
 * expansion computes as: y = 1/m refined by two Newton steps from v_rcp_f64, q = a * y,
 * r = fma(-m, q, a), result = fma(r, y, q) - wrapped in v_div_scale / v_div_fixup for operands near the
 * ends of the exponent range.  Here y is formed once and shared, and the wrapping is replaced by a range
 * check that sends anything unusual to the plain '/' (same value, slower).  Every result is the IEEE
 * quotient, so tuple_normalize's three divisions (:346) keep their bits. */
/* RT_FAST_DIV 1 builds the three routines below with their shared reciprocal; 0, the default, leaves every division to
 * the compiler's own expansion.  The shared form saves a third of the instructions of three divisions, but its range
 * check is a branch (two for each kind of operand), and since the kernels are built without machine-code
 * loop-invariant motion and no longer wait for spilled registers the branches cost more than the instructions save:
 * cube 12.8 -> 12.5 us, six spheres 77.0 -> 75.1 us, cube + spheres with mirrors 296 -> 286 us, lighthouse at 4K
 * 41.5 -> 41.0 us with all three plain (each of the three alone gives a part of it).  Both forms return the IEEE
 * quotient. */
#ifndef RT_FAST_DIV
#define RT_FAST_DIV 0
#endif
#ifndef RT_FAST_DIV3
#define RT_FAST_DIV3 RT_FAST_DIV
#endif
#ifndef RT_FAST_RECIP
#define RT_FAST_RECIP RT_FAST_DIV
#endif
#ifndef RT_FAST_DIV2
#define RT_FAST_DIV2 RT_FAST_DIV
#endif
__device__ __forceinline__ bool div_fast_ok(double v) {
  const double a = fabs(v);
  return a == 0.0 || (a > 1e-100 && a < 1e100);
}
__device__ __forceinline__ d3 div3_by(d3 a, double m) {
  if (RT_FAST_DIV3 && div_fast_ok(a.x) && div_fast_ok(a.y) && div_fast_ok(a.z) && m > 1e-100 && m < 1e100) {
    double y = __builtin_amdgcn_rcp(m);
    y = __builtin_fma(y, __builtin_fma(-m, y, 1.0), y);
    y = __builtin_fma(y, __builtin_fma(-m, y, 1.0), y);
    const double qx = a.x * y, qy = a.y * y, qz = a.z * y;
    return mk3(__builtin_fma(__builtin_fma(-m, qx, a.x), y, qx), __builtin_fma(__builtin_fma(-m, qy, a.y), y, qy),
               __builtin_fma(__builtin_fma(-m, qz, a.z), y, qz));
  }
  return mk3(a.x / m, a.y / m, a.z / m);
}
/* 1.0 / m, same construction (1 * y is exact, so the quotient estimate is y itself) */
__device__ __forceinline__ double recip_ieee(double m) {
  const double am = fabs(m);
  if (RT_FAST_RECIP && am > 1e-100 && am < 1e100) {
    double y = __builtin_amdgcn_rcp(m);
    y = __builtin_fma(y, __builtin_fma(-m, y, 1.0), y);
    y = __builtin_fma(y, __builtin_fma(-m, y, 1.0), y);
    return __builtin_fma(__builtin_fma(-m, y, 1.0), y, y);
  }
  return 1.0 / m;
}
/* two quotients by one divisor, same construction */
__device__ __forceinline__ void div2_by(double a, double b, double m, double& qa, double& qb) {
  const double am = fabs(m);
  if (RT_FAST_DIV2 && div_fast_ok(a) && div_fast_ok(b) && am > 1e-100 && am < 1e100) {
    double y = __builtin_amdgcn_rcp(m);
    y = __builtin_fma(y, __builtin_fma(-m, y, 1.0), y);
    y = __builtin_fma(y, __builtin_fma(-m, y, 1.0), y);
    const double q0 = a * y, q1 = b * y;
    qa = __builtin_fma(__builtin_fma(-m, q0, a), y, q0);
    qb = __builtin_fma(__builtin_fma(-m, q1, b), y, q1);
    return;
  }
  qa = a / m;
  qb = b / m;
}
__device__ __forceinline__ d3 normalize3(d3 a) {                                                   /* :344 */
  return div3_by(a, mag3(a));
}
__device__ __forceinline__ d3 reflect3(d3 in, d3 n) {                                              /* :1065 */
  const d3 two_n = scale3(n, 2.0);
  const double d = dot3(in, n);
  return sub3(in, scale3(two_n, d));
}

/* Experiment (A/B builds only, -DRT_SHADE_FMA=1): fused multiply-adds in lighting() and in the transform of a sphere's
 * normal - never in an intersection test.  The default build contracts nothing: the reference binary has no FMA. */
#ifndef RT_SHADE_FMA
#define RT_SHADE_FMA 0
#endif
__device__ __forceinline__ double dot3s(d3 a, d3 b) {
#if RT_SHADE_FMA
  return __builtin_fma(a.z, b.z, __builtin_fma(a.y, b.y, a.x * b.x));
#else
  return dot3(a, b);
#endif
}
__device__ __forceinline__ d3 reflect3s(d3 in, d3 n) {
#if RT_SHADE_FMA
  const double d = -2.0 * dot3s(in, n);
  return mk3(__builtin_fma(n.x, d, in.x), __builtin_fma(n.y, d, in.y), __builtin_fma(n.z, d, in.z));
#else
  return reflect3(in, n);
#endif
}

/* pow(x, y) as lighting() and schlick() call it (:1310 with y = material.shininess, 200 in every stock material
 * (:710); :1480 with y = 5).  pow() is the one operation on the path whose last bits already differ between
 * the reference's libm and any device library, and its device expansion is the largest block of instructions
 * of a shaded point (~150 against ~60 for everything else lighting() does).  For an integer-valued exponent
 * of at most 1024 the power is formed by squaring and multiplying instead: at most 10 squarings and 10
 * products, each rounded once, so the result is within y * 2^-53 relative of the exact power (2.3e-14 for
 * y = 200) - the same order as the difference between two correctly working pow() implementations after the
 * reference's own rounding of x, and four orders of magnitude inside what can move an 8-bit colour value.
 * Any other exponent goes to the library routine. */
#ifndef RT_POW_BY_SQUARING
#define RT_POW_BY_SQUARING 1 /* 0: always the library routine (A/B builds) */
#endif
__device__ __forceinline__ double rt_pow(double x, double y) {
  const int e = (int)y;
  if (RT_POW_BY_SQUARING && y >= 0.0 && y <= 1024.0 && (double)e == y) {
    double r = (e & 1) ? x : 1.0, b = x;
#pragma unroll 1
    for (int k = e >> 1; k != 0; k >>= 1) {
      b = b * b;
      if (k & 1) r = r * b;
    }
    return r;
  }
  return pow(x, y);
}

/* rows 0..2 of a row-major 4x4 whose bottom row is 0 0 0 1 (mat4x4_mul_tuple, :419) */
__device__ __forceinline__ d3 xform_point(const double* __restrict__ m, d3 p) { /* w == 1: 1 * m[3] == m[3] */
  return mk3(((p.x * m[0] + p.y * m[1]) + p.z * m[2]) + m[3], ((p.x * m[4] + p.y * m[5]) + p.z * m[6]) + m[7],
             ((p.x * m[8] + p.y * m[9]) + p.z * m[10]) + m[11]);
}
__device__ __forceinline__ d3 xform_vector(const double* __restrict__ m, d3 v) { /* w == 0 */
  return mk3((v.x * m[0] + v.y * m[1]) + v.z * m[2], (v.x * m[4] + v.y * m[5]) + v.z * m[6],
             (v.x * m[8] + v.y * m[9]) + v.z * m[10]);
}
/* transpose(m) * v with w == 0 (normal_at_shape_sphere_plane, :743-744) */
__device__ __forceinline__ d3 xform_vector_transposed(const double* __restrict__ m, d3 v) {
  return mk3((v.x * m[0] + v.y * m[4]) + v.z * m[8], (v.x * m[1] + v.y * m[5]) + v.z * m[9],
             (v.x * m[2] + v.y * m[6]) + v.z * m[10]);
}

/* the shading variants of the two transforms a sphere's normal goes through (see RT_SHADE_FMA) */
__device__ __forceinline__ d3 xform_point_s(const double* __restrict__ m, d3 p) {
#if RT_SHADE_FMA
  return mk3(__builtin_fma(p.z, m[2], __builtin_fma(p.y, m[1], __builtin_fma(p.x, m[0], m[3]))),
             __builtin_fma(p.z, m[6], __builtin_fma(p.y, m[5], __builtin_fma(p.x, m[4], m[7]))),
             __builtin_fma(p.z, m[10], __builtin_fma(p.y, m[9], __builtin_fma(p.x, m[8], m[11]))));
#else
  return xform_point(m, p);
#endif
}
__device__ __forceinline__ d3 xform_vector_transposed_s(const double* __restrict__ m, d3 v) {
#if RT_SHADE_FMA
  return mk3(__builtin_fma(v.z, m[8], __builtin_fma(v.y, m[4], v.x * m[0])), __builtin_fma(v.z, m[9], __builtin_fma(v.y, m[5], v.x * m[1])),
             __builtin_fma(v.z, m[10], __builtin_fma(v.y, m[6], v.x * m[2])));
#else
  return xform_vector_transposed(m, v);
#endif
}

#endif
