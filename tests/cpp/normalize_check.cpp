// Brute-force check that rtx::normalize (one shared division, raytracer_amd/csrc/rtmi_exact.h) returns the same
// bits as the reference's three divisions by the square root (Utilities.swift:27-30).  TEST INFRASTRUCTURE.
//   normalize_check <millions of vectors> [seed]
// Vectors are drawn from several families that stress the rounding of a quotient: random mantissas with
// exponents spread over a wide or a narrow range, mantissas of all ones or all zeros (powers of two), components
// that are tiny fractions of the length, axis-aligned and near-axis vectors (quotients just below 1, lengths whose
// significand is all ones: the cases a shared-division scheme is most likely to round wrongly), and vectors far outside the guard's
// range (which must take the plain divisions and still agree).  rtx::normalize3 (three at once) is held to
// rtx::normalize on every run of three consecutive vectors, which mixes guarded and unguarded ones.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../raytracer_amd/csrc/rtmi_exact.h"

static inline uint64_t next(uint64_t& s) {  // xorshift64*
  s ^= s >> 12; s ^= s << 25; s ^= s >> 27;
  return s * 0x2545F4914F6CDD1Dull;
}
static inline double from_bits(uint64_t b) { double d; memcpy(&d, &b, 8); return d; }
static inline uint64_t bits(double d) { uint64_t b; memcpy(&b, &d, 8); return b; }

static double draw(uint64_t& s, int family) {
  const uint64_t r = next(s);
  uint64_t mant = r & 0xFFFFFFFFFFFFFull;
  int e;
  switch (family) {
    case 0: e = (int)((r >> 52) % 41) - 20; break;              // everyday magnitudes
    case 1: e = (int)((r >> 52) % 1200) - 600; break;           // across and beyond the guard's range
    case 2: mant = 0xFFFFFFFFFFFFFull; e = (int)((r >> 52) % 9) - 4; break;          // all ones
    case 3: mant = 0; e = (int)((r >> 52) % 9) - 4; break;                            // powers of two
    case 4: mant = 0xFFFFFFFFFFFFFull - (r & 7); e = (int)((r >> 52) % 5) - 2; break; // just below a power of two
    case 5: mant = r & 15; e = (int)((r >> 52) % 5) - 2; break;                       // just above a power of two
    case 6: e = (int)((r >> 52) % 3) - 60 * (int)((r >> 60) & 1); break;              // one component tiny next to the others
    case 8: e = -24 - (int)((r >> 52) % 8); break;  // with a family-0 first component: a quotient just below 1 (significand near 2)
    case 9: e = -60; break;                          // with an all-ones first component: the length is that component, all ones
    default: e = (int)((r >> 52) % 3); break;
  }
  const uint64_t sign = (r >> 63) << 63;
  return from_bits(sign | ((uint64_t)(e + 1023) << 52) | mant);
}

int main(int argc, char** argv) {
  const long long millions = argc > 1 ? atoll(argv[1]) : 20;
  uint64_t seed = argc > 2 ? strtoull(argv[2], nullptr, 10) : 12345;
  long long bad = 0, fast = 0, bad3 = 0;
  const long long n = millions * 1000000LL;
#pragma omp parallel for reduction(+ : bad, fast, bad3) schedule(static)
  for (long long chunk = 0; chunk < n / 1000; ++chunk) {
    uint64_t s = (seed + 0x9E3779B97F4A7C15ull * (uint64_t)(chunk + 1)) | 1;
    rtx::d3 recent[3];  // the last three vectors: normalize3 must agree with normalize on every triple
    for (int k = 0; k < 1000; ++k) {
      const int family = (int)(next(s) % 10);
      rtx::d3 v = rtx::mk(draw(s, family), draw(s, family), draw(s, (next(s) & 3) ? family : 0));
      // directed cases: quotients whose significand is close to 2 (the first approximation of such a quotient may be two
      // ulps off), and a length whose significand is all ones
      if (family == 8) { v.x = draw(s, 0); v.y *= v.x; v.z = draw(s, 8) * v.x; }
      if (family == 9) { v.x = draw(s, 2); v.z = draw(s, 9) * v.x; }
      if ((next(s) & 31) == 0) v.y = 0.0;          // axis-aligned now and then
      if ((next(s) & 63) == 0) v.z = -0.0;
      const double length = sqrt(rtx::dp(v, v));
      const double ex = v.x / length, ey = v.y / length, ez = v.z / length;
      const rtx::d3 g = rtx::normalize(v);
      const bool same = (bits(g.x) == bits(ex) || (std::isnan(g.x) && std::isnan(ex))) &&
                        (bits(g.y) == bits(ey) || (std::isnan(g.y) && std::isnan(ey))) &&
                        (bits(g.z) == bits(ez) || (std::isnan(g.z) && std::isnan(ez)));
      if (!same) {
        if (bad < 5) fprintf(stderr, "mismatch: (%a, %a, %a) -> (%a, %a, %a), expected (%a, %a, %a)\n", v.x, v.y, v.z, g.x, g.y, g.z, ex, ey, ez);
        ++bad;
      }
      recent[k % 3] = v;
      if (k >= 2) {
        rtx::d3 out[3];
        rtx::normalize3(recent, out);
        for (int q = 0; q < 3; ++q) {
          const rtx::d3 one = rtx::normalize(recent[q]);
          if (memcmp(&one, &out[q], sizeof(one)) != 0 &&
              !(std::isnan(one.x) && std::isnan(out[q].x)) && !(std::isnan(one.y) && std::isnan(out[q].y)) && !(std::isnan(one.z) && std::isnan(out[q].z)))
            ++bad3;
        }
      }
      const double lo = 3.054936363499605e-151, hi = 3.273390607896142e+150;
      if (length >= lo && length <= hi && fabs(v.x) >= lo && fabs(v.y) >= lo && fabs(v.z) >= lo) ++fast;
    }
  }
  printf("normalize_check: %lld vectors, %lld through the shared division, %lld mismatches; normalize3 against normalize: %lld mismatches\n",
         n, fast, bad, bad3);
  return bad || bad3 ? 1 : 0;
}
