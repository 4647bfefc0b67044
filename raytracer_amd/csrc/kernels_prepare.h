// kernels_prepare.h -- part of rtmi.hip, which includes its parts in a fixed order (each relies on the ones before it).
// Scene preparation: triangle planes, the list scan's per-frame pack, the cone tree (rtmi_tree.h), the general-origin filter.
// Not a stand-alone header.
// ---------------------------------------------------------------------------------------
// scene preparation
// ---------------------------------------------------------------------------------------
__global__ void k_tri_planes(const rtmi_triangle* __restrict__ tris, long long n, TriPlane* planes) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const rtmi_triangle& t = tris[i];
  planes[i] = rtx::tri_plane(rtx::ld3(t.v[0]), rtx::ld3(t.v[1]), rtx::ld3(t.v[2]));
}

// Per frame, per triangle group: decide which triangles a ray from `cam` can hit at all
// (the reference's `t > 0`, Renderables.swift:150-152, is ray-independent for a common
// origin: t = num / -nddot with -nddot < 0, so t > 0 iff num < 0) and count them per block.
__global__ void k_pack_flags(const rtmi_triangle* __restrict__ tris, const TriPlane* __restrict__ planes,
                             const int* __restrict__ group, int n, d3 cam, unsigned char* keep,
                             int* block_counts) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  bool k = false;
  if (i < n) {
    const TriPlane pl = planes[group[i]];
    double num = pl.c - rtx::dp(rtx::mk(pl.nx, pl.ny, pl.nz), cam);
    k = num < 0;  // false for NaN (degenerate triangle): never hit in the reference either
    keep[i] = k;
  }
  unsigned long long m = __ballot(k);
  __shared__ int wsum[4];
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = __popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) block_counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// exclusive scan of the per-block counts, one workgroup
__global__ void k_pack_scan(int* block_counts, int nblocks, int* npacked_out) {
  __shared__ int part[1024];
  int t = threadIdx.x;
  int per = (nblocks + 1023) / 1024;
  int lo = t * per, hi = min(lo + per, nblocks);
  int s = 0;
  for (int i = lo; i < hi; ++i) s += block_counts[i];
  part[t] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    int v = t >= off ? part[t - off] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int run = part[t] - s;
  for (int i = lo; i < hi; ++i) {
    int c = block_counts[i];
    block_counts[i] = run;
    run += c;
  }
  if (t == 1023) *npacked_out = part[1023];
}

// Write the filter operands of every kept triangle at its compacted position.
//
// Two conservative FP32 tests are prepared per triangle, both for rays from the common
// origin o with unit direction d.
//
// (1) Cone test.  With a, b, c the unit vectors from o to the vertices, axis = normalize(a+b+c)
//     and cos_t = min(axis.a, axis.b, axis.c), every direction that hits the triangle satisfies
//     d.axis >= cos_t (the spherical triangle is the convex hull of its corners; valid while
//     cos_t > 0).  Stored: w = axis / (cos_t - 2^-19); nominate iff d.w >= 1.  The 2^-19 margin
//     exceeds the FP32 evaluation error of the dot product (< 8*2^-24*|axis|_1 <= 8.3e-7) plus
//     the FP64 rounding of axis/cos_t and of the reference.  A triangle with cos_t - 2^-19 < 1/16
//     (it subtends more than ~86 degrees) gets w = NaN and raises its batch's flag instead.
//
// (2) Edge test.  The reference's three inside tests (Renderables.swift:158-163) are
//     equivalent, given n.d < 0, to d.M_i <= 0 with M_0 = (v0-o)x(v1-o), M_1 = (v1-o)x(v2-o),
//     M_2 = (v2-o)x(v0-o): for the plane point P, e01 x (P - v0) = lambda*n and
//     d.(e01 x (P - v0)) = d.((v0-o)x(v1-o)).  Stored: M_i/E_i; nominate iff max_i d.(M_i/E_i) <= 1,
//     where E_i = 16*2^-24*|M_i|_1 + 2^-40*L^2 bounds the FP32 evaluation error (operand
//     roundings, one product and two fma roundings: < 8*2^-24*|M_i|_1 for |d_k| <= 1) plus the
//     rounding of the reference's own FP64 formulas (about 2^-46*L^2, L = |o|_1 + max|v|_1).
//
// Every pair the reference accepts passes both; pairs that pass both are re-decided in FP64.
__global__ void k_pack_write(const rtmi_triangle* __restrict__ tris, const int* __restrict__ group, int n,
                             d3 cam, const unsigned char* __restrict__ keep,
                             const int* __restrict__ block_offsets, float* cone, float* edge, int* packed_tri) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  bool k = i < n && keep[i];
  unsigned long long m = __ballot(k);
  __shared__ int wsum[4];
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) wsum[w] = __popcll(m);
  __syncthreads();
  if (!k) return;
  int pos = block_offsets[blockIdx.x] + __popcll(m & ((1ull << lane) - 1ull));
  for (int q = 0; q < w; ++q) pos += wsum[q];

  int ti = group[i];
  const rtmi_triangle& t = tris[ti];
  d3 A = rtx::sub(rtx::ld3(t.v[0]), cam), B = rtx::sub(rtx::ld3(t.v[1]), cam), Cc = rtx::sub(rtx::ld3(t.v[2]), cam);

  // (1) cone
  {
    double la = sqrt(rtx::dp(A, A)), lb = sqrt(rtx::dp(B, B)), lc = sqrt(rtx::dp(Cc, Cc));
    d3 a = rtx::divs(A, la), bb = rtx::divs(B, lb), c = rtx::divs(Cc, lc);
    d3 sum = rtx::add(rtx::add(a, bb), c);
    double ls = sqrt(rtx::dp(sum, sum));
    d3 axis = rtx::divs(sum, ls);
    double cos_t = fmin(rtx::dp(axis, a), fmin(rtx::dp(axis, bb), rtx::dp(axis, c)));
    double den = cos_t - 1.9073486328125e-06;  // 2^-19
    float wx, wy, wz;
    bool wide = !(den >= 0.0625);  // also true for NaN (a vertex at the origin, cancelling axes)
    if (!wide) {
      wx = (float)(axis.x / den); wy = (float)(axis.y / den); wz = (float)(axis.z / den);
      if (!isfinite(wx) || !isfinite(wy) || !isfinite(wz)) wide = true;
    }
    int b = pos / TRI_BATCH, j = pos % TRI_BATCH;
    float* dst = cone + (size_t)b * CONE_DWORDS + (j >> 1) * 6 + (j & 1);
    if (wide) {
      wx = wy = wz = __builtin_nanf("");
      reinterpret_cast<int*>(cone)[(size_t)b * CONE_DWORDS + CONE_FLAG] = 1;  // benign race: all writers store 1
    }
    dst[0] = wx; dst[2] = wy; dst[4] = wz;
  }

  // (2) edge planes
  d3 M[3] = {rtx::cross(A, B), rtx::cross(B, Cc), rtx::cross(Cc, A)};
  double L = fabs(cam.x) + fabs(cam.y) + fabs(cam.z);
  double lv = 0;
  for (int q = 0; q < 3; ++q) lv = fmax(lv, fabs(t.v[q][0]) + fabs(t.v[q][1]) + fabs(t.v[q][2]));
  L += lv;
  const double eps32 = 5.9604644775390625e-08;     // 2^-24
  const double abs_term = 9.094947017729282e-13;   // 2^-40
  float* dst = edge + (size_t)pos * EDGE_FLOATS;
  for (int e = 0; e < 3; ++e) {
    double E = 16.0 * eps32 * (fabs(M[e].x) + fabs(M[e].y) + fabs(M[e].z)) + abs_term * L * L;
    float mx = (float)(M[e].x / E), my = (float)(M[e].y / E), mz = (float)(M[e].z / E);
    if (!(E > 0) || !isfinite(mx) || !isfinite(my) || !isfinite(mz)) mx = my = mz = 0.0f;  // always nominate
    dst[e * 3 + 0] = mx;
    dst[e * 3 + 1] = my;
    dst[e * 3 + 2] = mz;
  }
  packed_tri[pos] = ti;
}

#include "rtmi_tree.h"

// General-origin filter data for the top-level triangle runs (rays that start on surfaces:
// reflection and shadow rays, RayTracingFunctions.swift:212,236).  All FP32 values are relative
// to the centre c0; r_o1 bounds |o - c0|_1 over every possible ray origin (scene bounds).
//
// (1) Sphere tests: the bounding-sphere tree over the runs' triangles (tree_sphere_records in
//     rtmi_tree.h, built on the host), walked by top_tree_trace below.
// (2) Plane and edge tests (Pluecker form), per nominated triangle, prepared here:
//       n.d <= Ea                        (Renderables.swift:146, front face)
//       n.o - cpl >= -Eb                 (t > 0, :150-152, given n.d < 0)
//       d.w_i + (o x d).e_i <= E_i       (:158-163; w_i = v_i x v_j, e_i = v_j - v_i, because
//                                         d.((v_i-o)x(v_j-o)) = d.(v_i x v_j) + (o x d).(v_j - v_i))
//     each bound 16*2^-24 times the sum of the magnitudes that enter the FP32 evaluation.
// Pairs that pass everything are decided by rtx::tri_hit in FP64.
__global__ void k_pack_general(const rtmi_triangle* __restrict__ tris, const int* __restrict__ group, int n, d3 c0,
                               double r_o1, float* gst2) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const rtmi_triangle& t = tris[group[i]];
  d3 v[3];
  for (int q = 0; q < 3; ++q) v[q] = rtx::sub(rtx::ld3(t.v[q]), c0);
  const double eps32 = 5.9604644775390625e-08;  // 2^-24
  d3 c = rtx::mk((v[0].x + v[1].x + v[2].x) / 3.0, (v[0].y + v[1].y + v[2].y) / 3.0, (v[0].z + v[1].z + v[2].z) / 3.0);
  double s1 = fabs(c.x) + fabs(c.y) + fabs(c.z) + r_o1;
  // (2)
  float* o = gst2 + (size_t)i * GST2_FLOATS;
  d3 e01 = rtx::sub(v[1], v[0]), e02 = rtx::sub(v[2], v[0]);
  d3 nn = rtx::normalize(rtx::cross(e01, e02));
  double cpl = rtx::dp(nn, v[0]);
  double n1 = fabs(nn.x) + fabs(nn.y) + fabs(nn.z);
  o[0] = (float)nn.x; o[1] = (float)nn.y; o[2] = (float)nn.z; o[3] = (float)cpl;
  o[4] = (float)(16.0 * eps32 * n1);                          // Ea
  o[5] = (float)(16.0 * eps32 * (n1 * r_o1 + fabs(cpl)));     // Eb
  for (int e = 0; e < 3; ++e) {
    d3 a = v[e], b2 = v[(e + 1) % 3];
    d3 w = rtx::cross(a, b2), ed = rtx::sub(b2, a);
    double E = 16.0 * eps32 * ((fabs(w.x) + fabs(w.y) + fabs(w.z)) + 2.0 * r_o1 * (fabs(ed.x) + fabs(ed.y) + fabs(ed.z))) +
               9.094947017729282e-13 * (s1 * s1);
    o[6 + e] = (float)E;
    float* q = o + 9 + e * 6;
    q[0] = (float)w.x; q[1] = (float)w.y; q[2] = (float)w.z;
    q[3] = (float)ed.x; q[4] = (float)ed.y; q[5] = (float)ed.z;
  }
  // a degenerate triangle (NaN normal) never hits in the reference; make it fail test (a)
  bool bad = false;
  for (int q = 0; q < 27; ++q) bad = bad || !isfinite(o[q]);
  if (bad) {
    for (int q = 0; q < 27; ++q) o[q] = 0.f;
    if (nn.x != nn.x || nn.y != nn.y || nn.z != nn.z) o[4] = -1.0f;  // n.d = 0 <= -1 is false: rejected
    // otherwise (overflow): zeros pass every test and the pair is decided in FP64
  }
}
