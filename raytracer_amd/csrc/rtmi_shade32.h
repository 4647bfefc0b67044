// rtmi_shade32.h -- calculateLighting (RayTraceUI/RayTracingFunctions.swift:182-271) for a triangle hit of a primary ray,
// evaluated in FP32 TOGETHER WITH A BOUND on how far each colour channel can be from the reference's FP64 value.
// Host and gfx950 device code.
//
// The only thing the reference does with a sample's colour is average the samples of a pixel and truncate
// 255 * mean to a byte (RTF:105-111, Utilities.swift:54-65).  A colour that is known to within 1e-5 decides that byte
// for all but the few pixels whose 255 * mean lies that close to a whole number; those pixels -- and every sample
// flagged here -- are evaluated again with the reference's own FP64 arithmetic (k_fix_pixels).  So the bytes the
// library returns are the reference's, and the FP64 lighting (seven square roots and divisions per hit with three
// lights) runs for about one pixel in a hundred.
//
// Scope: the case k_shade_candidates<INERT> serves -- the hit object is a triangle; nothing a reflection or shadow
// ray can hit changes a pixel (every member of the top-level list is a bounding sphere without a material), so the
// lighting is ambient + per light diffuse + specular, no shadow test; specular exponents are whole numbers 1..64.
//
// What is discontinuous in the reference's formulas is flagged instead of bounded: the sign test of
// normalLightVectorDp (RTF:229-232) decides whether a light contributes at all.  Everything else -- the clamps
// (RTF:263, :269), `reflectedDotViewVector <= 0` (RTF:255: the power tends to 0 there because the exponent is >= 1) --
// is continuous, and a continuous function's error follows from its argument's.
//
// Error model (u = 2^-24; all vectors below are unit vectors unless said otherwise; x* is the reference's value):
//   d        primary direction, |d - d*| <= dir_err (a per-frame constant: FP32 evaluation of
//            pixelToWorldCoordinate from the per-column and per-row tables, and of the normalisation)
//   n        float(n*): the triangle's unit normal, |n - n*| <= sqrt(3) u
//   num      float(num*), num* = c - n*.camera < 0 (Renderables.swift:150): |num - num*| <= u |num*|
//   nd       = -(n.d):  |nd - nd*| <= dir_err + 6u =: dn.   A hit with nd <= 8 dn is flagged (t unreliable).
//   t        = -num / nd (v_rcp_f32, 1 ulp): relative error rho_t <= dn / nd + 6u
//   Pc       = t d, the hit point relative to the camera: |Pc - Pc*| <= t (rho_t + dir_err + 2u) =: eP
//   Lv       = lc - Pc with lc = float(light.location - camera) (FP64 difference, rounded once):
//            |Lv - Lv*| <= u (|lc|_1 + 2 |Lv|) + eP =: eL
//   Lh       = Lv / |Lv| (v_rsq_f32, 1 ulp): |Lh - Lh*| <= 2 eL / |Lv| + 8u
//   ndl      = Lh.n: |ndl - ndl*| <= 2 eL / |Lv| + 14u =: dndl.   |ndl| <= dndl: the sample is flagged.
//   R        = 2 ndl n - Lh (the reference normalises it; its length is 1 to FP64 rounding): |R - R*| <= 3 dndl + 16u
//   rv       = R.V with V = -d (V* = normalize(camera - P*) = -d* to FP64 rounding): |rv - rv*| <= 3 dndl + dir_err + 24u =: drv
//   rv*^Ns   lies between max(rv - drv, 0)^Ns and (rv + drv)^Ns (binary powering, 14u relative each): the midpoint is
//            used, half the width plus the rounding is its error
// and the colour terms follow by multiplication with coefficients that are themselves rounded (3u relative).
// tests/cpp/shade32_check.cpp holds bound >= observed error on 10^8 random hits and reports how tight it is.
#pragma once

#include <math.h>

#include "rtmi_exact.h"

namespace rtx32 {

constexpr float U = 5.9604644775390625e-08f;  // 2^-24

#if defined(__HIP_DEVICE_COMPILE__)
RT_HD float rsq(float x) { return __builtin_amdgcn_rsqf(x); }  // v_rsq_f32: 1 ulp
RT_HD float rcp(float x) { return __builtin_amdgcn_rcpf(x); }  // v_rcp_f32: 1 ulp
#else
RT_HD float rsq(float x) { return 1.0f / sqrtf(x); }
RT_HD float rcp(float x) { return 1.0f / x; }
#endif

// Per frame and light: the values the FP32 lighting reads (built on the host in FP64, rounded once).
struct LightF {
  float lx, ly, lz, l1;  // light.location - camera, and the 1-norm of that vector
  float dr, dg, db, dmax;  // light.diffuse, and max |.| over the channels
  float sr, sg, sb, smax;  // light.specular
};

// Per frame and material.
struct MatF {
  float ar, ag, ab, amax;  // material.ambient * ambientLight (the FP64 product, rounded)
  float dr, dg, db, dmax;  // material.diffuse
  float sr, sg, sb, smax;  // material.specular
  int ns;                  // specular exponent, 1..64; 0: this material cannot be shaded here (every sample is flagged)
  int pad[3];
};

struct Shade32 {
  float r, g, b;  // the clamped colour; a channel holds exactly 1.0f ONLY IF the reference's value is exactly 1
  float err;      // every channel that is not exactly 1.0f lies within err of the reference's; +inf: flagged
};

// lo^n and hi^n for 0 <= lo <= hi and n >= 1 by binary powering, side by side (two products per packed instruction on
// gfx950): at most 12 roundings each for n <= 64.
RT_HD void powi2(float lo, float hi, int n, float* plo, float* phi) {
  float rl = 1.0f, rh = 1.0f, bl = lo, bh = hi;
  while (n) {
    if (n & 1) { rl = rl * bl; rh = rh * bh; }
    n >>= 1;
    if (n) { bl = bl * bl; bh = bh * bh; }
  }
  *plo = rl;
  *phi = rh;
}

// (a NaN may come out as 0 or 1 here: every NaN among the colour terms is one among the error terms too, and flags the sample)
#if defined(__HIP_DEVICE_COMPILE__)
RT_HD float clamp01f(float v) { return __builtin_amdgcn_fmed3f(v, 0.0f, 1.0f); }
#else
RT_HD float clamp01f(float v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); }
#endif
RT_HD float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

// d: unit primary direction (error dir_err); n, num: the triangle's plane as described above.
// UNIFORM_NS: the caller passes the exponent in ns_uniform and vouches that every material has that one (on the GPU the
// value then sits in a scalar register and the powering loop is scalar code: four packed multiplications for Ns = 8
// instead of four trips through a loop under an execution mask); else the material's own exponent is used.
template <bool UNIFORM_NS = false>
#if defined(__HIPCC__)
#define RT_HD_FLAT __host__ __device__ __forceinline__
#else
#define RT_HD_FLAT inline __attribute__((always_inline))
#endif
RT_HD_FLAT Shade32 shade_hit(float dx, float dy, float dz, float dir_err, float nx, float ny, float nz, float num, const MatF& m,
                             const LightF* lights, int n_lights, int ns_uniform = 0) {
  // (fused multiply-adds are written out: the translation unit is built without contraction, and a fused operation
  // only ever has less rounding error than the two the model above charges)
  const float inf = __builtin_inff();
  Shade32 out;
  bool flag = (UNIFORM_NS ? ns_uniform : m.ns) < 1;
  const float nd = -__builtin_fmaf(nz, dz, __builtin_fmaf(ny, dy, nx * dx));
  const float dn = dir_err + 6.0f * U;
  if (!(nd > 8.0f * dn)) flag = true;
  const float rnd = rcp(nd);
  const float t = -num * rnd;
  const float rho_t = __builtin_fmaf(dn, rnd, 6.0f * U);
  const float px = t * dx, py = t * dy, pz = t * dz;
  const float eP = fabsf(t) * (rho_t + (dir_err + 2.0f * U));
  float r = m.ar, g = m.ag, b = m.ab;  // RTF:193-196
  float mag = m.amax;                  // sum of the magnitudes of the terms added so far (largest channel)
  float er = U * m.amax;
#pragma nounroll
  for (int li = 0; li < n_lights; ++li) {
    const LightF L = lights[li];
    const float vx = L.lx - px, vy = L.ly - py, vz = L.lz - pz;
    const float ll = __builtin_fmaf(vz, vz, __builtin_fmaf(vy, vy, vx * vx));
    const float inv = rsq(ll);
    const float len = ll * inv;
    const float eL = __builtin_fmaf(U, __builtin_fmaf(2.0f, len, L.l1), eP);
    const float dndl = __builtin_fmaf(2.0f * eL, inv, 14.0f * U);
    const float hx = vx * inv, hy = vy * inv, hz = vz * inv;
    const float ndl = __builtin_fmaf(hz, nz, __builtin_fmaf(hy, ny, hx * nx));
    if (!(fabsf(ndl) > dndl)) flag = true;  // RTF:229-232 could go either way (also: NaN, a light at the hit point)
    if (!(ndl > 0.0f)) continue;
    // diffuse, RTF:246-249
    r = __builtin_fmaf(m.dr * L.dr, ndl, r); g = __builtin_fmaf(m.dg * L.dg, ndl, g); b = __builtin_fmaf(m.db * L.db, ndl, b);
    const float kdmax = m.dmax * L.dmax;
    mag = __builtin_fmaf(kdmax, ndl, mag);
    er = __builtin_fmaf(kdmax, __builtin_fmaf(6.0f * U, ndl, dndl), er);
    // specular, RTF:251-266
    const float two = 2.0f * ndl;
    const float rx = __builtin_fmaf(two, nx, -hx), ry = __builtin_fmaf(two, ny, -hy), rz = __builtin_fmaf(two, nz, -hz);
    const float rv = -__builtin_fmaf(rz, dz, __builtin_fmaf(ry, dy, rx * dx));
    const float drv = __builtin_fmaf(3.0f, dndl, dir_err + 24.0f * U);
    if (rv + drv > 0.0f) {
      // rv* lies in [rv - drv, rv + drv] and x -> max(x, 0)^Ns is monotonic: its value lies between the powers of the
      // two ends (each computed to 14u relative); the midpoint stands for it, half the width is its error
      float plo, phi;
      powi2(fmaxf(rv - drv, 0.0f), rv + drv, UNIFORM_NS ? ns_uniform : m.ns, &plo, &phi);
      const float spec = 0.5f * (plo + phi);
      const float espec = __builtin_fmaf(0.5f, phi - plo, 16.0f * U * phi);
      r += clamp01f(m.sr * L.sr * spec); g += clamp01f(m.sg * L.sg * spec); b += clamp01f(m.sb * L.sb * spec);  // RTF:259-266
      const float ksmax = m.smax * L.smax;
      mag += fminf(ksmax * spec, 1.0f);
      er = __builtin_fmaf(ksmax, __builtin_fmaf(6.0f * U, spec, espec), er);
    }
  }
  er = __builtin_fmaf((float)(2 * n_lights + 2) * U, mag, er);  // the roundings of the additions
  er = __builtin_fmaf(1.0e-9f, 1.0f + mag, er);                 // the reference's own FP64 roundings, many times over
  if (flag || !(er < 1.0f)) er = inf;                           // (also catches NaN)
  // RTF:269 clamps; a channel whose value is at least 1 whatever the error is exactly 1 in the reference too
  const float cr = clamp01f(r), cg = clamp01f(g), cb = clamp01f(b);
  const float below1 = 1.0f - U;  // the largest float below 1
  out.r = (r - er >= 1.0f) ? 1.0f : fminf(cr, below1);
  out.g = (g - er >= 1.0f) ? 1.0f : fminf(cg, below1);
  out.b = (b - er >= 1.0f) ? 1.0f : fminf(cb, below1);
  out.err = er + U;  // (the step from 1 to the float below it)
  if (!(r == r) || !(g == g) || !(b == b)) out.err = inf;
  return out;
}

}  // namespace rtx32
