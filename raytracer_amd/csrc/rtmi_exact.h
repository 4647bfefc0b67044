// rtmi_exact.h -- FP64 evaluation of the reference's intersection and shading arithmetic,
// operation for operation, for host and gfx950 device code.
//
// These functions decide every result the library returns: the FP32 packed filter in
// rtmi.hip only nominates (ray, triangle) pairs, and each nominee is re-decided here with
// the reference's own formulas, operand order and divisions.  The translation unit must be
// built with -ffp-contract=off (the reference never fuses a*b+c); sqrt and / on gfx950
// double are correctly rounded, so the bits match a host evaluation.
//
// Citations: RTF = RayTraceUI/RayTracingFunctions.swift, REN = RayTraceUI/Renderables.swift,
// WCS = RayTraceUI/WorldCoordinateSequence.swift, UTL = RayTraceUI/Utilities.swift.
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define RT_HD __host__ __device__ inline
#else
#define RT_HD inline
#endif

namespace rtx {

struct d3 {
  double x, y, z;
};

RT_HD d3 mk(double x, double y, double z) { d3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_HD d3 ld3(const double* p) { return mk(p[0], p[1], p[2]); }
RT_HD d3 add(d3 a, d3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_HD d3 sub(d3 a, d3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_HD d3 neg(d3 a) { return mk(-a.x, -a.y, -a.z); }
RT_HD d3 smul(double s, d3 a) { return mk(s * a.x, s * a.y, s * a.z); }  // scalar * vector
RT_HD d3 muls(d3 a, double s) { return mk(a.x * s, a.y * s, a.z * s); }  // vector * scalar
RT_HD d3 divs(d3 a, double s) { return mk(a.x / s, a.y / s, a.z / s); }  // vector / scalar

// dp, UTL:23-25
RT_HD double dp(d3 a, d3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// lenSquared, UTL:38-40
RT_HD double len_squared(d3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
// normalize, UTL:27-30: three true divisions by the square root.  The three quotients share one division.
// With r = RN(1/length) (an IEEE division, so within half an ulp of 1/length) a quotient x/length is built in
// three steps:
//   q0 = RN(x*r)                        relative error below 2^-52, i.e. up to about two ulps when the quotient's
//                                       significand is close to 2 -- not yet a faithful rounding;
//   q1 = RN(q0 + RN(x - q0*length)*r)   the residual is computed by one fma with a relative error of at most 2^-53,
//                                       so q0 + residual*r lies within 2^-50 ulp of x/length before its rounding:
//                                       q1 is within 0.5 + 2^-50 ulp of x/length, a faithful rounding;
//   q2 = RN(q1 + (x - q1*length)*r)     Markstein's correction step (Markstein 1990; Muller et al., Handbook of
//                                       Floating-Point Arithmetic, "correctly rounded division"): for a faithful q1
//                                       the residual x - q1*length is exactly representable (the fma returns it
//                                       without rounding), and with r within half an ulp of 1/length the result is
//                                       RN(x/length), the bits of the plain division, in every rounding case.
// (Stopping at q1 -- as this function did at first -- gives the same bits unless x/length lies within 2^-50 ulp of the
// midpoint of two doubles, which random testing cannot reach; q2 rests on the theorem instead.)  All of it holds
// while nothing under- or overflows on the way, which the guard ensures: length and every component within
// 2^-500 .. 2^500 in magnitude (a zero component takes the plain divisions too, they keep its sign).  On gfx950
// this is one division and fifteen multiply-adds instead of three divisions (an IEEE division is a 13-instruction
// sequence there).
RT_HD double shared_quotient(double x, double length, double r) {
  const double q0 = x * r;
  const double q1 = __builtin_fma(__builtin_fma(-q0, length, x), r, q0);
  return __builtin_fma(__builtin_fma(-q1, length, x), r, q1);
}
RT_HD d3 normalize(d3 a) {
  const double length = sqrt(dp(a, a));
  const double lo = 3.054936363499605e-151, hi = 3.273390607896142e+150;  // 2^-500, 2^500
  const double ax = fabs(a.x), ay = fabs(a.y), az = fabs(a.z);
  if (length >= lo && length <= hi && ax >= lo && ay >= lo && az >= lo) {  // components cannot exceed length by more than an ulp
    const double r = 1.0 / length;
    return mk(shared_quotient(a.x, length, r), shared_quotient(a.y, length, r), shared_quotient(a.z, length, r));
  }
  return mk(a.x / length, a.y / length, a.z / length);
}
// N normalizations at once, each the same bits as normalize(): the square roots and the divisions are written side
// by side, without a branch between them, so that their instruction sequences -- each a chain of a dozen dependent
// operations -- can be issued interleaved.  The guard is checked once, afterwards; in the rare case that it fails
// for any of them, all are redone by normalize().
template <int N>
RT_HD void normalize_n(const d3 (&a)[N], d3 (&out)[N]) {
  const double lo = 3.054936363499605e-151, hi = 3.273390607896142e+150;  // 2^-500, 2^500
  double length[N], r[N];
  bool ok = true;
  for (int k = 0; k < N; ++k) length[k] = sqrt(dp(a[k], a[k]));
  for (int k = 0; k < N; ++k) r[k] = 1.0 / length[k];
  for (int k = 0; k < N; ++k) {
    out[k] = mk(shared_quotient(a[k].x, length[k], r[k]), shared_quotient(a[k].y, length[k], r[k]), shared_quotient(a[k].z, length[k], r[k]));
    ok = ok && length[k] >= lo && length[k] <= hi && fabs(a[k].x) >= lo && fabs(a[k].y) >= lo && fabs(a[k].z) >= lo;
  }
  if (!ok) {
    for (int k = 0; k < N; ++k) out[k] = normalize(a[k]);
  }
}
RT_HD void normalize3(const d3 (&a)[3], d3 (&out)[3]) { normalize_n<3>(a, out); }
// cross, UTL:32-36
RT_HD d3 cross(d3 a, d3 b) {
  return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// clampFunction(0, 1), UTL:67-78
RT_HD double clamp01(double v) {
  if (v < 0.0) return 0.0;
  if (v > 1.0) return 1.0;
  return v;
}

// The ray-independent head of Triangle.intersections (REN:137-143): unit normal and plane
// constant.  Evaluated once per triangle; the bits equal a per-call evaluation.
struct TriPlane {
  double nx, ny, nz, c;
};

RT_HD TriPlane tri_plane(d3 v0, d3 v1, d3 v2) {
  d3 v0v1 = sub(v1, v0);
  d3 v0v2 = sub(v2, v0);
  d3 n = normalize(cross(v0v1, v0v2));
  TriPlane p;
  p.nx = n.x; p.ny = n.y; p.nz = n.z;
  p.c = dp(n, v0);
  return p;
}

// The rest of Triangle.intersections (REN:144-166).  Returns true on a hit and then sets
// *t and *P (the normal of the hit is the plane normal).
RT_HD bool tri_hit(d3 v0, d3 v1, d3 v2, const TriPlane& pl, d3 origin, d3 direction, double* t_out,
                   d3* P_out) {
  d3 normal = mk(pl.nx, pl.ny, pl.nz);
  double nddot = dp(normal, neg(direction));
  if (!(nddot > 0)) return false;  // REN:146 `nddot <= 0 -> return`; NaN falls through there
                                   // and fails `t > 0` below, so no hit either way
  double t = (pl.c - dp(normal, origin)) / -nddot;
  if (t > 0) {
    d3 v0v1 = sub(v1, v0);
    d3 v0v2 = sub(v2, v0);
    d3 v1v2 = sub(v2, v1);
    d3 v2v0 = neg(v0v2);
    d3 point = add(origin, smul(t, direction));
    d3 v0p = sub(point, v0);
    d3 v1p = sub(point, v1);
    d3 v2p = sub(point, v2);
    d3 a = cross(v0v1, v0p);
    d3 b = cross(v1v2, v1p);
    d3 c = cross(v2v0, v2p);
    if (dp(a, normal) >= 0 && dp(b, normal) >= 0 && dp(c, normal) >= 0) {
      *t_out = t;
      *P_out = point;
      return true;
    }
  }
  return false;
}

// The intersection parameter alone (REN:144-152), for a pair that tri_hit has already accepted:
// the same two operations, hence the same value.
RT_HD double tri_t(const TriPlane& pl, d3 origin, d3 direction) {
  d3 normal = mk(pl.nx, pl.ny, pl.nz);
  double nddot = dp(normal, neg(direction));
  return (pl.c - dp(normal, origin)) / -nddot;
}

// Sphere.intersections (REN:248-282) reduced to its first element: the smallest root that
// passes the >= 1e-7 filter, with its point and the un-normalised outward vector p - location
// (REN:276 normalises it; callers that never read the normal skip that).
RT_HD bool sphere_first_raw(d3 location, double radius, d3 origin, d3 direction, double* t_out,
                            d3* P_out, d3* raw_n_out) {
  double radius_squared = radius * radius;  // pow(radius, 2), REN:210/:243
  d3 center_to_eye = sub(origin, location);
  double a = -dp(direction, center_to_eye);
  double delta = a * a - (len_squared(center_to_eye) - radius_squared);  // pow(a, 2)
  if (!(delta >= 0)) return false;  // REN:255 `delta < 0 -> return`; NaN: sqrt(NaN) roots fail the filter
  double sqrtdelta = sqrt(delta);
  double r0, r1;
  int nr;
  if (delta == 0) {
    r0 = a; r1 = a; nr = 1;
  } else {
    double t1 = a + sqrtdelta, t2 = a - sqrtdelta;
    if (t1 < t2) { r0 = t1; r1 = t2; } else { r0 = t2; r1 = t1; }
    nr = 2;
  }
  double t;
  if (r0 >= 0.0000001) t = r0;
  else if (nr == 2 && r1 >= 0.0000001) t = r1;
  else return false;
  d3 p = add(origin, smul(t, direction));
  *t_out = t;
  *P_out = p;
  *raw_n_out = sub(p, location);
  return true;
}

RT_HD bool sphere_first(d3 location, double radius, d3 origin, d3 direction, double* t_out,
                        d3* P_out, d3* n_out) {
  d3 raw;
  if (!sphere_first_raw(location, radius, origin, direction, t_out, P_out, &raw)) return false;
  *n_out = normalize(raw);
  return true;
}

// Whether Sphere.intersections appends anything (used by shadow rays, RTF:236-243).
RT_HD bool sphere_any(d3 location, double radius, d3 origin, d3 direction) {
  double radius_squared = radius * radius;
  d3 center_to_eye = sub(origin, location);
  double a = -dp(direction, center_to_eye);
  double delta = a * a - (len_squared(center_to_eye) - radius_squared);
  if (!(delta >= 0)) return false;
  double sqrtdelta = sqrt(delta);
  if (delta == 0) return a >= 0.0000001;
  double t1 = a + sqrtdelta, t2 = a - sqrtdelta;
  return t1 >= 0.0000001 || t2 >= 0.0000001;
}

// The same answer for callers that can afford a branch (ray generation): t1 = a + sqrt(delta) is at least a
// (the root is not negative and rounding is monotonic), so with a itself past the threshold -- the sphere's
// centre lies ahead of the origin -- no square root is needed.
RT_HD bool sphere_any_ahead(d3 location, double radius, d3 origin, d3 direction) {
  double radius_squared = radius * radius;
  d3 center_to_eye = sub(origin, location);
  double a = -dp(direction, center_to_eye);
  double delta = a * a - (len_squared(center_to_eye) - radius_squared);
  if (!(delta >= 0)) return false;
  if (delta == 0 || a >= 0.0000001) return a >= 0.0000001;
  double sqrtdelta = sqrt(delta);
  double t1 = a + sqrtdelta, t2 = a - sqrtdelta;
  return t1 >= 0.0000001 || t2 >= 0.0000001;
}

// PointLight.intersections (REN:80-91) with eq3 (UTL:155-157).
// The three quotients can only be equal if num and direction are parallel to within rounding: tx == ty makes
// num.x * direction.y and num.y * direction.x agree to 2^-52 relative, and a pair of products further apart than
// 1e-12 relative (the slack covers the rounding of the products and of their difference many times over) rules the
// hit out without a division.  A NaN or an overflow in the products fails the comparison and the divisions decide.
RT_HD bool light_hit(d3 location, d3 origin, d3 direction, double* t_out, d3* P_out) {
  d3 num = sub(location, origin);
  {
    const double pxy = num.x * direction.y, pyx = num.y * direction.x;
    const double pyz = num.y * direction.z, pzy = num.z * direction.y;
    const double sxy = fabs(pxy) + fabs(pyx), syz = fabs(pyz) + fabs(pzy);
    // (products near the subnormal range carry no such relative accuracy: those go to the divisions too)
    if ((sxy > 1e-290 && fabs(pxy - pyx) > 1e-12 * sxy) || (syz > 1e-290 && fabs(pyz - pzy) > 1e-12 * syz)) return false;
  }
  double tx = num.x / direction.x, ty = num.y / direction.y, tz = num.z / direction.z;
  if (tx >= 0.0000001 && (!isinf(tx) && tx == ty && ty == tz)) {
    *t_out = tx;
    *P_out = add(origin, smul(tx, direction));
    return true;
  }
  return false;
}

// getPlaneVectors (RTF:46-57).
RT_HD void plane_vectors(d3 direction, d3* u_out, d3* v_out) {
  d3 plane_normal = normalize(neg(direction));
  d3 up = normalize(mk(0.0, 1.0, -1.0));
  d3 u = normalize(cross(up, plane_normal));
  *u_out = u;
  *v_out = cross(plane_normal, u);
}

// pixelToWorldCoordinate (WCS:62-66).
RT_HD d3 pixel_to_world(d3 ul, d3 u, d3 v, double ppw_h, double ppw_v, double i, double j) {
  d3 uc = divs(muls(u, i), ppw_h);
  d3 vc = divs(muls(v, j), ppw_v);
  return sub(add(ul, uc), vc);
}

}  // namespace rtx
