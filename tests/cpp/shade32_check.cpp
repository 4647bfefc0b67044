// Brute-force check of the error bound of the FP32 lighting (raytracer_amd/csrc/rtmi_shade32.h) against the reference's
// FP64 lighting for the same hit (RayTracingFunctions.swift:182-271 without secondary rays, as rtx:: states it).
// TEST INFRASTRUCTURE.
//   shade32_check <millions of hits> [seed]
// A hit is drawn at random: camera, unit direction, a plane that the ray meets from its front at a parameter between
// 0.05 and 200, one to four lights anywhere from 0.01 to 200 away from the hit point, a material with any colours and a
// whole-number exponent.  The FP32 side gets what the kernels give it: the direction rounded to float and moved by up
// to dir_err, the FP64 unit normal and plane constant rounded to float, the lights relative to the camera rounded to
// float.  For every hit that the FP32 lighting does not flag:
//   * each channel must lie within Shade32::err of the FP64 value;
//   * a channel that holds exactly 1.0f must be exactly 1 in FP64.
// Reports the share of flagged hits and how much of the bound the worst observed error used.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../raytracer_amd/csrc/rtmi_shade32.h"

using rtx::d3;

static inline uint64_t next(uint64_t& s) {  // xorshift64*
  s ^= s >> 12; s ^= s << 25; s ^= s >> 27;
  return s * 0x2545F4914F6CDD1Dull;
}
static inline double unif(uint64_t& s) { return (double)(next(s) >> 11) * (1.0 / 9007199254740992.0); }
static inline double range(uint64_t& s, double lo, double hi) { return lo + (hi - lo) * unif(s); }
static inline double logrange(uint64_t& s, double lo, double hi) { return lo * exp(unif(s) * log(hi / lo)); }
static d3 unit_vec(uint64_t& s) {
  for (;;) {
    d3 v = rtx::mk(range(s, -1, 1), range(s, -1, 1), range(s, -1, 1));
    double l2 = rtx::dp(v, v);
    if (l2 > 1e-4 && l2 <= 1.0) return rtx::normalize(v);
  }
}

struct Light64 { d3 loc; double spec[3], diff[3]; };
struct Mat64 { double Ns, Kd[3], Ka[3], Ks[3]; };

// calculateLighting for a triangle hit when no secondary ray can change the result (RTF:189-202, :226-270).
static void lighting64(d3 P, d3 n, d3 camera, const Mat64& m, const double amb[3], const Light64* L, int nl, double out[3]) {
  double I[3] = {m.Ka[0] * amb[0], m.Ka[1] * amb[1], m.Ka[2] * amb[2]};
  d3 V = rtx::normalize(rtx::sub(camera, P));
  for (int li = 0; li < nl; ++li) {
    d3 ptl = rtx::normalize(rtx::sub(L[li].loc, P));
    double ndl = rtx::dp(ptl, n);
    if (ndl <= 0) continue;
    for (int c = 0; c < 3; ++c) I[c] = I[c] + m.Kd[c] * (ndl * L[li].diff[c]);
    d3 lr = rtx::normalize(rtx::sub(rtx::smul(2 * ndl, n), ptl));
    double rv = rtx::dp(lr, V);
    if (rv <= 0) continue;
    double st = pow(rv, m.Ns);
    for (int c = 0; c < 3; ++c) I[c] = I[c] + rtx::clamp01(m.Ks[c] * (st * L[li].spec[c]));
  }
  for (int c = 0; c < 3; ++c) out[c] = rtx::clamp01(I[c]);
}

int main(int argc, char** argv) {
  const long long millions = argc > 1 ? atoll(argv[1]) : 20;
  const uint64_t seed = argc > 2 ? strtoull(argv[2], nullptr, 10) : 2026;
  const long long n = millions * 1000000LL;
  long long bad = 0, flagged = 0, sat_checked = 0, bad_sat = 0;
  double worst_ratio = 0.0, sum_err = 0.0, max_err_unflagged = 0.0;
#pragma omp parallel for reduction(+ : bad, flagged, sat_checked, bad_sat, sum_err) reduction(max : worst_ratio, max_err_unflagged) schedule(static)
  for (long long chunk = 0; chunk < n / 1000; ++chunk) {
    uint64_t s = (seed + 0x9E3779B97F4A7C15ull * (uint64_t)(chunk + 1)) | 1;
    for (int k = 0; k < 1000; ++k) {
      const double scale = logrange(s, 0.1, 1000.0);  // how far from the origin the scene lives
      const d3 camera = rtx::mk(range(s, -scale, scale), range(s, -scale, scale), range(s, -scale, scale));
      const d3 d = unit_vec(s);
      // a plane the ray meets from its front: -(n.d) between grazing and head-on
      d3 nrm;
      double nd;
      do { nrm = unit_vec(s); nd = -rtx::dp(nrm, d); } while (!(nd > ((next(s) & 7) ? 0.02 : 1e-5)));
      const double t0 = logrange(s, 0.05, 200.0);
      const d3 Q = rtx::add(camera, rtx::smul(t0, d));
      rtx::TriPlane pl;
      pl.nx = nrm.x; pl.ny = nrm.y; pl.nz = nrm.z; pl.c = rtx::dp(nrm, Q);
      const double num = pl.c - rtx::dp(nrm, camera);
      if (!(num < 0)) continue;
      const double t = rtx::tri_t(pl, camera, d);
      const d3 P = rtx::add(camera, rtx::smul(t, d));
      const int nl = 1 + (int)(next(s) % 4);
      Light64 L[4];
      for (int li = 0; li < nl; ++li) {
        L[li].loc = rtx::add(P, rtx::smul(logrange(s, 0.01, 200.0), unit_vec(s)));
        for (int c = 0; c < 3; ++c) { L[li].spec[c] = range(s, 0, 1.2); L[li].diff[c] = range(s, 0, 1.2); }
      }
      Mat64 m;
      static const int ns_choices[8] = {1, 2, 3, 8, 8, 16, 33, 64};
      m.Ns = ns_choices[next(s) & 7];
      const bool grey = next(s) & 1;
      for (int c = 0; c < 3; ++c) {
        m.Kd[c] = grey && c ? m.Kd[0] : range(s, 0, 1);
        m.Ka[c] = grey && c ? m.Ka[0] : range(s, 0, 1);
        m.Ks[c] = grey && c ? m.Ks[0] : range(s, 0, 1.5);
      }
      const double amb[3] = {range(s, 0, 0.5), range(s, 0, 0.5), range(s, 0, 0.5)};
      double ref[3];
      lighting64(P, nrm, camera, m, amb, L, nl, ref);

      // what the kernel hands to the FP32 lighting
      const float dir_err = (float)logrange(s, 1e-7, 8e-6);
      const d3 wob = rtx::smul(0.8 * dir_err * unif(s), unit_vec(s));
      const float dx = (float)(d.x + wob.x), dy = (float)(d.y + wob.y), dz = (float)(d.z + wob.z);
      rtx32::LightF LF[4];
      for (int li = 0; li < nl; ++li) {
        const d3 lc = rtx::sub(L[li].loc, camera);
        LF[li].lx = (float)lc.x; LF[li].ly = (float)lc.y; LF[li].lz = (float)lc.z;
        LF[li].l1 = (float)(fabs(lc.x) + fabs(lc.y) + fabs(lc.z)) * 1.0000002f;
        LF[li].dr = (float)L[li].diff[0]; LF[li].dg = (float)L[li].diff[1]; LF[li].db = (float)L[li].diff[2];
        LF[li].dmax = rtx32::max3f(fabsf(LF[li].dr), fabsf(LF[li].dg), fabsf(LF[li].db));
        LF[li].sr = (float)L[li].spec[0]; LF[li].sg = (float)L[li].spec[1]; LF[li].sb = (float)L[li].spec[2];
        LF[li].smax = rtx32::max3f(fabsf(LF[li].sr), fabsf(LF[li].sg), fabsf(LF[li].sb));
      }
      rtx32::MatF MF;
      MF.ar = (float)(m.Ka[0] * amb[0]); MF.ag = (float)(m.Ka[1] * amb[1]); MF.ab = (float)(m.Ka[2] * amb[2]);
      MF.amax = rtx32::max3f(fabsf(MF.ar), fabsf(MF.ag), fabsf(MF.ab));
      MF.dr = (float)m.Kd[0]; MF.dg = (float)m.Kd[1]; MF.db = (float)m.Kd[2];
      MF.dmax = rtx32::max3f(fabsf(MF.dr), fabsf(MF.dg), fabsf(MF.db));
      MF.sr = (float)m.Ks[0]; MF.sg = (float)m.Ks[1]; MF.sb = (float)m.Ks[2];
      MF.smax = rtx32::max3f(fabsf(MF.sr), fabsf(MF.sg), fabsf(MF.sb));
      MF.ns = (int)m.Ns;
      const rtx32::Shade32 o = (next(s) & 1) ? rtx32::shade_hit<true>(dx, dy, dz, dir_err, (float)nrm.x, (float)nrm.y, (float)nrm.z, (float)num, MF, LF, nl, MF.ns)
                                             : rtx32::shade_hit<false>(dx, dy, dz, dir_err, (float)nrm.x, (float)nrm.y, (float)nrm.z, (float)num, MF, LF, nl);
      if (!(o.err < 1.0f)) { ++flagged; continue; }
      const float got[3] = {o.r, o.g, o.b};
      for (int c = 0; c < 3; ++c) {
        if (got[c] == 1.0f) {
          ++sat_checked;
          if (ref[c] != 1.0) { ++bad_sat; if (bad_sat < 5) fprintf(stderr, "channel says exactly 1, reference %.17g\n", ref[c]); }
          continue;
        }
        const double e = fabs((double)got[c] - ref[c]);
        const double ratio = e / (double)o.err;
        if (ratio > worst_ratio) worst_ratio = ratio;
        if (e > (double)o.err) {
          if (bad < 5) fprintf(stderr, "error %.3g above bound %.3g (channel %d, Ns %d, lights %d, nd %.3g, t %.3g, dir_err %.3g)\n", e, (double)o.err, c, MF.ns, nl, nd, t, (double)dir_err);
          ++bad;
        }
      }
      sum_err += (double)o.err;
      if ((double)o.err > max_err_unflagged) max_err_unflagged = (double)o.err;
    }
  }
  printf("shade32_check: %lld hits, %lld flagged (%.3f %%), %lld errors above their bound, %lld of %lld saturated channels not exactly 1; "
         "worst error used %.3f of its bound; mean bound %.3g, largest %.3g\n",
         n, flagged, 100.0 * (double)flagged / (double)n, bad, bad_sat, sat_checked, worst_ratio, sum_err / (double)(n - flagged), max_err_unflagged);
  return bad || bad_sat ? 1 : 0;
}
