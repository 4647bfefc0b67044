// rtmi.hip -- gfx950 (MI355X) kernels and the C ABI of include/rtmi.h.
//
// Replaces the reference's raytraceWorld / raytracePixels / calculateLighting call tree
// (RayTraceUI/RayTracingFunctions.swift:59-114, :145-271) with a staged GPU pipeline:
//
//   k_frame_tables, k_pack_tree   per camera: sample tables; cones of the triangles and of the
//                                 tree nodes above them (rtmi_tree.h)
//   k_gen               per sample: FP32 pre-test, ray generation, sphere/light tests (FP64)
//   k_search_tree       per wave of 256 rays: one walk of the cone tree, packed-FP32 cone and
//                       edge tests; emits (sample, list position) candidates
//   k_search_exact      per candidate: the reference's triangle test in FP64; lowest list
//                       position per ray
//   k_shade_candidates  per candidate: the ray's intersection is shaded (calculateLighting, FP64)
//   k_shade             per queued hit (spheres, lights, top-level triangles): the same shading
//   k_resolve           per pixel: sample average, x255 truncation, RGBA8 store
//
// RTMI_SEARCH_LIST_SCAN replaces the three search kernels by k_search_primary (k_pack_flags /
// _scan / _write): every front-facing triangle in list order until the ray's first hit, one lane
// per 4 rays, triangle operands wave-uniform in SGPRs, two triangles per v_pk_fma_f32.
// Scenes that list triangles at top level add k_search_finish / k_resolve_top (primary rays) and
// walk a bounding-sphere tree for their reflection and shadow rays (top_tree_trace).
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (see Makefile).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/rtmi.h"
#include "rtmi_exact.h"

using rtx::d3;
using rtx::TriPlane;

typedef float f2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------
static thread_local std::string g_last_error;

void rtmi_set_last_error(const std::string& msg) { g_last_error = msg; }  // used by rtmi_obj.cpp

#define HIP_TRY(expr)                                                                      \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess) {                                                                \
      g_last_error = std::string(#expr) + ": " + hipGetErrorString(e_) + " (" + __FILE__ + \
                     ":" + std::to_string(__LINE__) + ")";                                 \
      return e_ == hipErrorOutOfMemory ? RTMI_ERR_OUT_OF_MEMORY : RTMI_ERR_HIP;            \
    }                                                                                      \
  } while (0)

// ---------------------------------------------------------------------------------------
// device-side constants and helpers
// ---------------------------------------------------------------------------------------
constexpr int BLK = 8;              // pixel block edge; samples are enumerated block by block
constexpr int BLK_PIX = BLK * BLK;  // so that 64 consecutive samples are screen neighbours
constexpr int TRI_BATCH = 8;        // triangles per scalar-load batch of the cone loop
constexpr int CONE_DWORDS = 32;     // per batch: 4 pairs x 3 components x (tri a, tri b) + flag + pad
constexpr int CONE_FLAG = 24;       // dword index of the batch's "always look closer" flag
constexpr int EDGE_FLOATS = 9;      // per triangle: 3 scaled edge-plane vectors
constexpr int GST2_FLOATS = 32;     // per triangle: plane (4) + bounds (5) + 3 x (w, e) (18), padded

struct FrameConst {
  double ul[3], u[3], v[3], cam[3];
  double ppw_h, ppw_v, sub;
  double ambient[3];
  int W, H, x0, y0, x1, y1;
  int spp, n1;  // samples per pixel = n1*n1 (WorldCoordinateSequence.swift:74-78)
  int spp_shift, n1_shift;  // log2 of the two when they are powers of two, else -1
  float inv_spp;
  const double* utab;  // per sample column: (u * i) / pixelsPerWorldHorizontal
  const double* vtab;  // per sample row:    (v * j) / pixelsPerWorldVertical
  // FP32 pre-test of k_gen: the same tables rounded to FP32, ul - cam, and one cone per sphere-like
  // object (unit vector towards its centre, and the squared cosine a direction must reach to
  // come anywhere near it); n_genf == 0 switches the pre-test off
  const float* utab_f;
  const float* vtab_f;
  const float4* genf;
  float base_f[3];
  int n_genf;
  int tile_size, tiles_x, tile_rank, tile_world;
};

// Work counters are striped over NSLOT cache lines (one 64-byte line per slot) so that the
// per-wave atomic adds of a launch do not serialise on one address; the host sums the slots.
enum { C_REFL = 0, C_SHADOW, C_EXACT, C_HITS, C_TESTS_PRIMARY, C_TESTS_SECONDARY, C_SHADOW_ELIDED, C_NODE_TESTS, C_NCOUNTERS = 8 };
constexpr int NSLOT = 64;

struct DevCounters {
  unsigned long long slot[NSLOT][C_NCOUNTERS];
  int q_child, q_shade, q_top;  // queue fill levels
  int overflow;                 // a candidate list of the tree search did not fit
  int bug;                      // nonzero: an index check failed (see checked()); the call reports an error
  int pad[3];
  unsigned long long cand_total;  // candidates of all tree search launches of the frame
  // The tree search keeps NSLOT candidate lists (a wave appends to list `wave & 63`), each with its
  // fill level on a cache line of its own: returning atomics on ONE address complete at roughly
  // one per 30 ns, which a launch of 200,000 waves cannot afford.
  // Two sets, used alternately by successive search launches of a frame: the last kernel of a
  // launch clears the set of the launch before it.
  int cand_n[2][NSLOT][16];
};

struct SceneDev {
  const rtmi_object* objects;
  const rtmi_light* lights;
  const rtmi_material* materials;
  const rtmi_triangle* tris;
  const TriPlane* planes;
  int n_objects, n_lights, n_sphere_like;
  int n_tris, n_samples;  // entries of tris/planes; entries of the per-sample arrays (for checked())
  long long n_top_tris;
  // general-origin filter of the top-level triangle runs (secondary rays), built once per scene
  const int* top_group;  // top-level ordinal -> triangle index
  const int* tri_owner;  // triangle -> list index of the object (run) that lists it
  double c0[3];          // centre the FP32 filter coordinates are relative to
};

// Pixel `pix` (0..63, row-major in the 8x8 block) of block `b`; false if it is not rendered.
__device__ inline bool pixel_coords(const FrameConst& fc, const int2* __restrict__ blocks, int b, int pix,
                                    int* x, int* y) {
  int2 bc = blocks[b];
  int px = bc.x * BLK + (pix & (BLK - 1));
  int py = bc.y * BLK + (pix >> 3);
  *x = px;
  *y = py;
  if (px < fc.x0 || px >= fc.x1 || py < fc.y0 || py >= fc.y1) return false;
  if (fc.tile_size > 0 && fc.tile_world > 1) {
    int tile = (py / fc.tile_size) * fc.tiles_x + (px / fc.tile_size);
    if (tile % fc.tile_world != fc.tile_rank) return false;
  }
  return true;
}

// Sample index within a batch: s = ((b*spp + k) << 6) + pix.  The 64 lanes of a wave are the 64
// pixels of one 8x8 block at the same sub-sample k, so per-sample arrays are read and written
// coalesced by k_gen, the searches and k_resolve alike, and everything derived from t = s >> 6
// (block, sub-sample, its column and row offsets) is wave-uniform scalar arithmetic.
__device__ inline void group_coords(const FrameConst& fc, int t, int* b_out, int* kx_out, int* ky_out) {
  int b, k;
  if (fc.spp_shift >= 0) {  // 1, 4, 16 ... samples per pixel
    b = t >> fc.spp_shift;
    k = t & (fc.spp - 1);
  } else {
    b = (int)((float)t * fc.inv_spp);  // t / spp for t < 2^22, corrected below
    k = t - b * fc.spp;
    if (k < 0) { --b; k += fc.spp; }
    if (k >= fc.spp) { ++b; k -= fc.spp; }
  }
  int kx;
  if (fc.n1_shift >= 0) kx = k >> fc.n1_shift; else kx = k / fc.n1;
  *b_out = b;
  *kx_out = kx;               // x outer, y inner (WorldCoordinateSequence.swift:74-78)
  *ky_out = k - kx * fc.n1;
}

// Primary ray direction of sub-sample (kx, ky) of pixel (x, y): WorldCoordinateSequence.swift:62-78
// and RayTracingFunctions.swift:73.  The two quotients of pixelToWorldCoordinate depend only on
// the sample's column and row; k_frame_tables evaluates them once per frame.
__device__ inline d3 sample_dir(const FrameConst& fc, int x, int y, int kx, int ky) {
  const double* uc = fc.utab + (size_t)(x * fc.n1 + kx) * 3;
  const double* vc = fc.vtab + (size_t)(y * fc.n1 + ky) * 3;
  d3 p = rtx::sub(rtx::add(rtx::ld3(fc.ul), rtx::ld3(uc)), rtx::ld3(vc));  // ul + uComponent - vComponent
  return rtx::normalize(rtx::sub(p, rtx::ld3(fc.cam)));
}

// uComponent = (u * i) / pixelsPerWorldHorizontal for every sample column i = x + kx*sub, and
// vComponent = (v * j) / pixelsPerWorldVertical for every sample row (WCS:63-64, :74-75).
__global__ void k_frame_tables(FrameConst fc, double* utab, double* vtab, float* utab_f, float* vtab_f) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int nu = fc.W * fc.n1, nv = fc.H * fc.n1;
  if (i < nu) {
    int x = i / fc.n1, kx = i - x * fc.n1;
    double c = fma((double)kx, fc.sub, (double)x);  // stride(from:to:by:) = start.addingProduct(k, by)
    d3 r = rtx::divs(rtx::muls(rtx::ld3(fc.u), c), fc.ppw_h);
    utab[(size_t)i * 3 + 0] = r.x; utab[(size_t)i * 3 + 1] = r.y; utab[(size_t)i * 3 + 2] = r.z;
    utab_f[(size_t)i * 3 + 0] = (float)r.x; utab_f[(size_t)i * 3 + 1] = (float)r.y; utab_f[(size_t)i * 3 + 2] = (float)r.z;
  } else if (i < nu + nv) {
    int j = i - nu;
    int y = j / fc.n1, ky = j - y * fc.n1;
    double c = fma((double)ky, fc.sub, (double)y);
    d3 r = rtx::divs(rtx::muls(rtx::ld3(fc.v), c), fc.ppw_v);
    vtab[(size_t)j * 3 + 0] = r.x; vtab[(size_t)j * 3 + 1] = r.y; vtab[(size_t)j * 3 + 2] = r.z;
    vtab_f[(size_t)j * 3 + 0] = (float)r.x; vtab_f[(size_t)j * 3 + 1] = (float)r.y; vtab_f[(size_t)j * 3 + 2] = (float)r.z;
  }
}

// An index that was READ FROM DEVICE MEMORY and is about to become an address.  Only a defect
// elsewhere can put it out of range; if one does, the index is replaced by 0 and the call ends with
// an error (DevCounters::bug names the check) instead of a memory fault on the GPU.
__device__ __forceinline__ int checked(int v, int n, DevCounters* cnt, int code) {
  if (__builtin_expect((unsigned)v >= (unsigned)n, 0)) {
    cnt->bug = code;
    return 0;
  }
  return v;
}

// Sum v over the wave and add it to this wave's stripe of counter `idx`.
__device__ inline void wave_count(DevCounters* cnt, int idx, unsigned long long v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
  if ((threadIdx.x & 63) == 0 && v) {
    int slot = (blockIdx.x * 4 + (threadIdx.x >> 6)) & (NSLOT - 1);
    atomicAdd(&cnt->slot[slot][idx], v);
  }
}

// Append `value` of every thread with `pred` to a global queue with ONE returning atomic per
// 256-thread workgroup.  Must be reached by all threads of the workgroup.
__device__ inline void block_append(int* counter, int* queue, bool pred, int value) {
  __shared__ int wtot[4];
  __shared__ int gbase;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  unsigned long long mask = __ballot(pred);
  if (lane == 0) wtot[w] = __popcll(mask);
  __syncthreads();
  if (threadIdx.x == 0) {
    int tot = wtot[0] + wtot[1] + wtot[2] + wtot[3];
    gbase = tot ? atomicAdd(counter, tot) : 0;
  }
  __syncthreads();
  if (pred) {
    int pos = gbase + __popcll(mask & ((1ull << lane) - 1ull));
    for (int q = 0; q < w; ++q) pos += wtot[q];
    queue[pos] = value;
  }
  __syncthreads();
}

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

// ---------------------------------------------------------------------------------------
// top-level tests against spheres, bounding spheres and lights (FP64, list order)
// ---------------------------------------------------------------------------------------
struct Hit {
  d3 P, n;        // n is the unit normal, or (n_unit == 0) the un-normalised P - centre of a sphere
  double t;
  int has_normal;
  int n_unit;
  int material;
  int id;  // triangle index, or -2 - object index
};

// Index of the first sphere-like member of the top-level list (sphere, bounding sphere or
// light) that the ray intersects at all, skipping triangle runs (they are searched
// separately); -1 if none.  Only the index is needed here: a bounding sphere's own hit point is
// never used (RayTracingFunctions.swift:85-93) and shading recomputes a sphere's.
__device__ inline int first_sphere_like(const SceneDev& sc, d3 o, d3 d) {
  for (int k = 0; k < sc.n_objects; ++k) {
    const rtmi_object& ob = sc.objects[k];
    if (ob.kind == RTMI_SPHERE || ob.kind == RTMI_BOUNDING_SPHERE) {
      if (rtx::sphere_any(rtx::ld3(ob.center), ob.radius, o, d)) return k;
    } else if (ob.kind == RTMI_LIGHT) {
      double t;
      d3 P;
      if (rtx::light_hit(rtx::ld3(sc.lights[ob.first].loc), o, d, &t, &P)) return k;
    }
  }
  return -1;
}

// The bounding-sphere tree of the top-level triangle runs as the shading kernels see it.  The
// three arrays are separate __restrict__ kernel arguments (that is what lets the compiler fetch
// records through scalar loads in kernels that also store).
struct TopTreeRef {
  const f2* rec;        // sphere records, all levels (tree_sphere_records)
  const float* gst2;    // per top-level ordinal: plane and edge operands (k_pack_general)
  const int* slot_pos;  // slot -> top-level ordinal (= position among the top-level triangles)
  int ltop, n_root, n_slots, n_records;
  int off[TREE_MAX_LEVELS];
};

// The scalar part of TopTreeRef, passed to the kernels by value.
struct TopTreeArgs {
  int ltop, n_root, n_slots, n_records;
  int off[TREE_MAX_LEVELS];
};

__device__ inline TopTreeRef make_top_tree(const TopTreeArgs& a, const f2* rec, const float* gst2, const int* slot_pos) {
  TopTreeRef t;
  t.rec = rec; t.gst2 = gst2; t.slot_pos = slot_pos;
  t.ltop = a.ltop; t.n_root = a.n_root; t.n_slots = a.n_slots; t.n_records = a.n_records;
#pragma unroll
  for (int l = 0; l < TREE_MAX_LEVELS; ++l) t.off[l] = a.off[l];
  return t;
}

// A general-origin ray (reflection or shadow ray) against ALL top-level triangles: the ordinal of
// the hit that comes first in list order (any_only: of some hit), or -1.  The lanes that call this
// together walk the tree together, like k_search_tree: a record of eight bounding spheres is
// evaluated for every calling lane's own ray (packed line-sphere test, two spheres per
// instruction), and a child is entered if some lane passes.  At a leaf the per-triangle cascade
// of k_pack_general follows, then rtx::tri_hit in FP64.  Called from divergent shading code: the
// wave-level votes only see the lanes that are tracing.
__device__ inline int top_tree_trace(const SceneDev& sc, const TopTreeRef& T, d3 o, d3 d, bool any_only, double* t_out,
                                     d3* P_out, unsigned long long* n_tests) {
  const float ox = (float)(o.x - sc.c0[0]), oy = (float)(o.y - sc.c0[1]), oz = (float)(o.z - sc.c0[2]);
  const float dx = (float)d.x, dy = (float)d.y, dz = (float)d.z;
  const float ndo = -(dx * ox + dy * oy + dz * oz);
  const float noo = -(ox * ox + oy * oy + oz * oz);
  const float o2x = 2.0f * ox, o2y = 2.0f * oy, o2z = 2.0f * oz;
  const float mx_ = oy * dz - oz * dy, my_ = oz * dx - ox * dz, mz_ = ox * dy - oy * dx;  // o x d for the edge tests
  const f2 X = {dx, dx}, Y = {dy, dy}, Z = {dz, dz}, N0 = {ndo, ndo};
  const f2 OX = {o2x, o2x}, OY = {o2y, o2y}, OZ = {o2z, o2z}, N1 = {noo, noo};
  int best = 0x7fffffff;  // ordinal of the best hit so far
  unsigned int leaf_visits = 0;

  // line-sphere values of the record's pair `pr`: passes iff !(U < 0)
  auto pair_u = [&](const f2* rp, int pr) {
    const f2 cx = rp[pr * 4 + 0], cy = rp[pr * 4 + 1], cz = rp[pr * 4 + 2], nk = rp[pr * 4 + 3];
    f2 A = __builtin_elementwise_fma(cz, Z, __builtin_elementwise_fma(cy, Y, __builtin_elementwise_fma(cx, X, N0)));
    f2 B = __builtin_elementwise_fma(cz, OZ, __builtin_elementwise_fma(cy, OY, __builtin_elementwise_fma(cx, OX, N1)));
    return __builtin_elementwise_fma(A, A, B) + nk;
  };
  auto level_off = [&](int lvl) {
    return lvl == 1 ? T.off[1] : lvl == 2 ? T.off[2] : lvl == 3 ? T.off[3] : lvl == 4 ? T.off[4] : lvl == 5 ? T.off[5]
         : lvl == 6 ? T.off[6] : T.off[7];
  };
  auto eval = [&](int lvl, int ri) -> unsigned {
    const f2* rp = T.rec + (size_t)min(level_off(lvl) + ri, T.n_records - 1) * (REC_DWORDS / 2);  // the clamp never binds
    const bool live = !any_only || best == 0x7fffffff;
    unsigned mask = 0;
#pragma unroll
    for (int pr = 0; pr < 4; ++pr) {
      f2 U = pair_u(rp, pr);
      if (__any(live && !(U.x < 0.0f))) mask |= 1u << (2 * pr);
      if (__any(live && !(U.y < 0.0f))) mask |= 2u << (2 * pr);
    }
    return (unsigned)__builtin_amdgcn_readfirstlane((int)mask);
  };
  auto leaf = [&](int ri) {
    if ((unsigned)(ri * 8) >= (unsigned)T.n_slots) return;  // cannot happen (absent nodes never pass); keeps a bad record from becoming a bad address
    const f2* rp = T.rec + (size_t)(T.off[0] + ri) * (REC_DWORDS / 2);
    const bool live = !any_only || best == 0x7fffffff;
    if (live) ++leaf_visits;
#pragma unroll
    for (int pr = 0; pr < 4; ++pr) {
      f2 U = pair_u(rp, pr);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float u = h ? U.y : U.x;
        bool cand = live && !(u < 0.0f);
        if (!__any(cand)) continue;
        const int g = T.slot_pos[ri * 8 + pr * 2 + h];  // top-level ordinal
        if ((unsigned)g >= (unsigned)sc.n_top_tris) continue;  // padding slot (its sphere passes nothing; belt and braces)
        cand = cand && (any_only ? best == 0x7fffffff : g < best);  // a later list position cannot replace the hit
        if (!__any(cand)) continue;
        const float* s2 = T.gst2 + (size_t)g * GST2_FLOATS;
        {
          const float nx = s2[0], ny = s2[1], nz = s2[2], cpl = s2[3], Ea = s2[4], Eb = s2[5];
          float nd = __builtin_fmaf(nz, dz, __builtin_fmaf(ny, dy, nx * dx));
          float no = __builtin_fmaf(nz, oz, __builtin_fmaf(ny, oy, nx * ox)) - cpl;
          cand = cand && (nd <= Ea) && (no >= -Eb);
        }
        if (!__any(cand)) continue;
#pragma unroll
        for (int e = 0; e < 3; ++e) {
          const float* w = s2 + 9 + e * 6;
          float sv = __builtin_fmaf(w[2], dz, __builtin_fmaf(w[1], dy, w[0] * dx));
          sv = __builtin_fmaf(w[5], mz_, __builtin_fmaf(w[4], my_, __builtin_fmaf(w[3], mx_, sv)));
          cand = cand && (sv <= s2[6 + e]);
        }
        if (!__any(cand)) continue;
        if (cand) {
          const int ti = min(max(sc.top_group[g], 0), sc.n_tris - 1);  // the clamps never bind
          const rtmi_triangle& tr = sc.tris[ti];
          double t;
          d3 P;
          if (rtx::tri_hit(rtx::ld3(tr.v[0]), rtx::ld3(tr.v[1]), rtx::ld3(tr.v[2]), sc.planes[ti], o, d, &t, &P)) {
            best = g;
            *t_out = t;
            *P_out = P;
          }
        }
      }
    }
  };

  const int ltop = T.ltop;
  bool done = false;
  for (int t = 0; t < T.n_root && !done; ++t) {
    if (ltop == 0) {
      leaf(t);
      done = any_only && !__any(best == 0x7fffffff);
      continue;
    }
    int lvl = ltop, ri = t;
    unsigned long long masks = (unsigned long long)eval(lvl, ri) << (8 * lvl);
    for (;;) {
      const unsigned cur = (unsigned)(masks >> (8 * lvl)) & 0xffu;
      if (cur == 0) {
        if (lvl == ltop) break;
        ++lvl;
        ri >>= 3;
        continue;
      }
      const int c = __builtin_ctz(cur);
      masks &= ~(1ull << (8 * lvl + c));
      const int child = ri * 8 + c;
      if (lvl == 1) {
        leaf(child);
        if (any_only && !__any(best == 0x7fffffff)) { done = true; break; }  // every tracing lane has its hit
      } else {
        --lvl;
        ri = child;
        masks |= (unsigned long long)eval(lvl, ri) << (8 * lvl);
      }
    }
  }
  *n_tests += 8ull * leaf_visits;  // (ray, triangle) sphere tests this lane's ray took part in
  return best == 0x7fffffff ? -1 : best;
}

// traceRay over the whole top-level list (RayTracingFunctions.swift:37-44) for a secondary
// ray: intersections[0] if want_first, and whether any non-bounding object is hit.
template <bool HAS_TOP>
__device__ inline bool trace_top_secondary(const SceneDev& sc, const TopTreeRef& tt, d3 o, d3 d, bool want_first, Hit* h,
                                           bool* any_non_bounding, unsigned long long* n_tests) {
  bool found = false, anb = false;
  // spheres, bounding spheres and lights first, in list order: for a shadow ray the first
  // non-bounding hit settles the question and the triangles need not be searched
  int first_obj = 0x7fffffff;  // list index of the first sphere-like hit (want_first)
#pragma nounroll
  for (int k = 0; k < sc.n_objects; ++k) {
    if (want_first ? found : anb) break;
    const rtmi_object& ob = sc.objects[k];
    if (ob.kind == RTMI_SPHERE || ob.kind == RTMI_BOUNDING_SPHERE) {
      double t;
      d3 P, n;
      if (want_first) {
        if (rtx::sphere_first_raw(rtx::ld3(ob.center), ob.radius, o, d, &t, &P, &n)) {
          found = true;
          first_obj = k;
          // without a material only the SIGN of normal.light is ever read (calculate_lighting)
          h->n_unit = ob.material >= 0;
          h->P = P; h->n = h->n_unit ? rtx::normalize(n) : n; h->t = t; h->has_normal = 1;
          h->material = ob.material; h->id = -2 - k;
        }
      } else if (ob.kind == RTMI_SPHERE) {
        if (rtx::sphere_any(rtx::ld3(ob.center), ob.radius, o, d)) anb = true;
      }
    } else if (ob.kind == RTMI_LIGHT) {
      double t;
      d3 P;
      if (rtx::light_hit(rtx::ld3(sc.lights[ob.first].loc), o, d, &t, &P)) {
        anb = true;
        if (want_first) {
          found = true;
          first_obj = k;
          h->P = P; h->t = t; h->has_normal = 0; h->n_unit = 1; h->material = -1; h->id = -2 - k;
        }
      }
    }
  }
  if constexpr (HAS_TOP) {
    // the top-level triangles: one walk for all runs; a triangle hit wins over the sphere-like
    // hit if its run is listed earlier (RTF:41-43: the first intersection in list order)
    if (want_first || !anb) {
      double t;
      d3 P;
      const int g = top_tree_trace(sc, tt, o, d, !want_first, &t, &P, n_tests);
      if (g >= 0) {
        anb = true;
        const int ti = min(max(sc.top_group[g], 0), sc.n_tris - 1);
        if (want_first && sc.tri_owner[ti] < first_obj) {
          found = true;
          const TriPlane pl = sc.planes[ti];
          h->P = P; h->n = rtx::mk(pl.nx, pl.ny, pl.nz); h->t = t; h->has_normal = 1; h->n_unit = 1;
          h->material = sc.tris[ti].material; h->id = ti;
        }
      }
    }
  }
  *any_non_bounding = anb;
  return found;
}

// ---------------------------------------------------------------------------------------
// k_gen: one thread per sample of the batch
// ---------------------------------------------------------------------------------------
// Outcome per sample (hit_id): -1 miss / unrendered; >= 0 triangle; <= -2 sphere-like object.
// With no top-level triangle runs the sample is resolved here: a sphere or light hit goes to
// the shade queue, a bounding-sphere hit to the child-search queue.  With top-level runs
// every rendered sample goes on to the top-level triangle search, which resolves it.
constexpr int GEN_ITERS = 8;  // samples per thread in k_gen; a workgroup covers 2048 consecutive samples

__global__ void __launch_bounds__(256)
k_gen(FrameConst fc, SceneDev sc, const int2* __restrict__ blocks, int nsamples, int* hit_id, double* dirs,
      int* q_child, int* q_shade, int* q_top, DevCounters* cnt) {
  // queue entries are staged in LDS and flushed with one global atomic per queue per workgroup
  __shared__ int lq[3][256 * GEN_ITERS];
  __shared__ int ln[3];
  __shared__ int lbase[3];
  if (threadIdx.x < 3) ln[threadIdx.x] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const bool has_top = sc.n_top_tris > 0;
  for (int it = 0; it < GEN_ITERS; ++it) {
    const int t = (blockIdx.x * GEN_ITERS + it) * 4 + wv;  // wave-uniform: the 64-sample group of this wave
    const int s = (t << 6) + lane;
    int x = 0, y = 0, kx = 0, ky = 0;
    bool live = false;
    if ((t << 6) < nsamples) {  // nsamples is a multiple of 64
      int b;
      group_coords(fc, t, &b, &kx, &ky);
      live = pixel_coords(fc, blocks, b, lane, &x, &y);
    }
    int id = -1;
    bool to_child = false, to_shade = false;
    if (live && fc.n_genf > 0) {
      // FP32 pre-test: a sample whose direction stays clear of every object's cone hits nothing,
      // and none of the FP64 work below is needed for it (hit_id = -1, no direction stored)
      const float* uc = fc.utab_f + (size_t)(x * fc.n1 + kx) * 3;
      const float* vc = fc.vtab_f + (size_t)(y * fc.n1 + ky) * 3;
      const float fx = fc.base_f[0] + uc[0] - vc[0], fy = fc.base_f[1] + uc[1] - vc[1], fz = fc.base_f[2] + uc[2] - vc[2];
      const float dd = fx * fx + fy * fy + fz * fz;
      bool maybe = false;
      for (int o = 0; o < fc.n_genf && !maybe; ++o) {
        const float4 gf = fc.genf[o];
        const float qd = gf.x * fx + gf.y * fy + gf.z * fz;
        maybe = gf.w <= 0.0f || (qd > 0.0f && !(qd * qd < gf.w * dd));
      }
      live = maybe;
    }
    if (live) {
      d3 d = sample_dir(fc, x, y, kx, ky);
      int obj = first_sphere_like(sc, rtx::ld3(fc.cam), d);
      if (obj >= 0) {
        id = -2 - obj;
        if (sc.objects[obj].kind == RTMI_BOUNDING_SPHERE) to_child = true; else to_shade = true;
      }
      if (obj >= 0 || has_top) {  // kept for the FP64 decisions of the search and for shading
        dirs[(size_t)s * 3 + 0] = d.x;
        dirs[(size_t)s * 3 + 1] = d.y;
        dirs[(size_t)s * 3 + 2] = d.z;
      }
    }
    if (s < nsamples) hit_id[s] = id;
    // with top-level triangle runs every live sample is resolved after their search
    bool pred[3] = {!has_top && to_child, !has_top && to_shade, has_top && live};
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      unsigned long long mask = __ballot(pred[q]);
      if (mask) {
        int leader = __ffsll((long long)mask) - 1;
        int base = 0;
        if (lane == leader) base = atomicAdd(&ln[q], __popcll(mask));
        base = __shfl(base, leader);
        if (pred[q]) lq[q][base + __popcll(mask & ((1ull << lane) - 1ull))] = s;
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    int* gcount = threadIdx.x == 0 ? &cnt->q_child : (threadIdx.x == 1 ? &cnt->q_shade : &cnt->q_top);
    lbase[threadIdx.x] = ln[threadIdx.x] ? atomicAdd(gcount, ln[threadIdx.x]) : 0;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    int* gq = q == 0 ? q_child : (q == 1 ? q_shade : q_top);
    for (int i = threadIdx.x; i < ln[q]; i += 256) gq[lbase[q] + i] = lq[q][i];
  }
}

// After the top-level triangle search: combine the first sphere-like object (hit_id from
// k_gen) with the first top-level triangle (tri_hit) by list position.
__global__ void __launch_bounds__(256)
k_resolve_top(SceneDev sc, const int* __restrict__ q_top, const int* __restrict__ nq_ptr, const int* __restrict__ tri_hit,
              const int* __restrict__ tri_owner, int* hit_id, int* q_child, int* q_shade, DevCounters* cnt) {
  int q = blockIdx.x * blockDim.x + threadIdx.x;
  bool to_child = false, to_shade = false;
  int s = -1;
  if (q < *nq_ptr) {
    s = checked(q_top[q], sc.n_samples, cnt, 14);
    int id = hit_id[s];
    int th = tri_hit[q];
    if (th >= 0) th = checked(th, sc.n_tris, cnt, 15);
    int obj = id <= -2 ? -2 - id : 0x7fffffff;
    if (th >= 0 && tri_owner[th] < obj) { id = th; obj = -1; }
    hit_id[s] = id;
    if (id >= 0) to_shade = true;
    else if (id <= -2) {
      if (sc.objects[checked(obj, sc.n_objects, cnt, 16)].kind == RTMI_BOUNDING_SPHERE) to_child = true; else to_shade = true;
    }
  }
  block_append(&cnt->q_child, q_child, to_child, s);
  block_append(&cnt->q_shade, q_shade, to_shade, s);
}

// ---------------------------------------------------------------------------------------
// k_search_primary: the list scan (RTMI_SEARCH_LIST_SCAN)
// ---------------------------------------------------------------------------------------
// Each lane owns R rays (queue entries base + r*64 + lane) that share the camera origin.
// Every (ray, triangle) pair goes through a cascade of three tests, cheapest first:
//
//   cone   FP32, packed: d.w >= 1            3 FMA per pair   (all pairs)
//   edge   FP32:         max_i d.m_i <= 1    9 FMA per pair   (pairs that pass the cone)
//   exact  FP64, the reference's own arithmetic (rtx::tri_hit)  (pairs that pass both)
//
// Cone operands arrive eight triangles at a time as 32 wave-uniform dwords, so the compiler
// keeps them in SGPRs (two s_load_dwordx16, next batch requested while the current one is
// evaluated) and feeds them to v_pk_mul_f32 / v_pk_fma_f32 as the scalar operand, two
// triangles per instruction, the ray component broadcast by op_sel.  Steady state per
// (ray, triangle): 1.5 packed FMA-class instructions and half a v_max3_f32; no LDS, no MFMA,
// no cross-lane traffic.  When some lane's running maximum reaches 1 (or the batch is
// flagged) the wave takes the rare path for that batch: per triangle, cone again (to find
// which), then the edge test, then the FP64 decision for the nominated rays, in list order.
// A ray that gets its hit is poisoned with NaN (it nominates nothing afterwards); the wave
// leaves the loop once every one of its rays has a hit (list-order policy: the first hit is
// final).
//
// mode 0: queue entries are sample ids awaiting the children of bounding object
//         `obj_filter`; the result is written to hit_id[s] and hits are appended to q_shade.
// mode 1: top-level runs; the result (triangle index or -1) goes to tri_hit[q].
// (Both modes apply to the cone tree's kernels as well.)

// Everything a search kernel needs besides its operand stream, one block per triangle group.  It
// lives in device memory and is read where it is used (prologue, rare path, epilogue), so that the
// list scan's steady-state loop keeps its scalar registers for the two triangle batches in flight.
struct SearchArgs {
  const int* queue;
  const int* nq_ptr;
  const int* npacked_ptr;
  const float* edge;
  const int* packed_tri;
  const double* dirs;
  double* tval;
  int* hit_id;
  int* tri_hit;
  int* q_shade;
  DevCounters* cnt;
  const rtmi_triangle* tris;
  const TriPlane* planes;
  double cam[3];
  int mode, obj_filter;
  const int* group;  // list position -> triangle index (the tree search keeps positions)
  int n_group;       // triangles in the group (entries of `group`)
  int n_samples, n_tris, n_records;  // entries of the per-sample arrays, of tris/planes, of the tree's records
  int* best_pos;     // per sample: lowest list position hit so far (tree search)
  int2* cand;        // (sample, list position) candidates of the tree search, NSLOT lists of cand_cap / NSLOT
  int cand_cap;
  TreeDev tree;
};

// The R rays of a lane: queue entries base + r*64 + lane.  An entry that does not belong to this
// launch (past the queue end, or a sample whose first sphere-like hit is another object) gets
// sid = -1 and a NaN direction, which passes no FP32 test.
template <int R>
__device__ __forceinline__ void search_prologue(const SearchArgs* ap, int base, int nq, int lane, int (&sid)[R],
                                                float (&dx)[R], float (&dy)[R], float (&dz)[R]) {
  const float qnan = __builtin_nanf("");
  const int* queue = ap->queue;
  const int* hit_id = ap->hit_id;
  const double* dirs = ap->dirs;
  const int mode = ap->mode, obj_filter = ap->obj_filter;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    int q = base + r * 64 + lane;
    bool valid = q < nq;
    sid[r] = valid ? checked(queue[q], ap->n_samples, ap->cnt, 1) : -1;
    if (valid && mode == 0 && hit_id[sid[r]] != -2 - obj_filter) sid[r] = -1;
    dx[r] = dy[r] = dz[r] = qnan;
    if (sid[r] >= 0) {
      const double* dp = dirs + (size_t)sid[r] * 3;
      dx[r] = (float)dp[0]; dy[r] = (float)dp[1]; dz[r] = (float)dp[2];
    }
  }
}

// best[r]: triangle index of ray r's hit, or -1.  Must be reached by every thread of the workgroup.
template <int R>
__device__ __forceinline__ void search_epilogue(const SearchArgs* ap, int base, int nq, int lane, const int (&sid)[R],
                                                const int (&best)[R]) {
  if (ap->mode == 0) {
    // hits go to the shade queue: one returning atomic per workgroup for all R rays
    int* hit_id = ap->hit_id;
    int* q_shade = ap->q_shade;
    DevCounters* cnt = ap->cnt;
    __shared__ int wtot[4];
    __shared__ int gbase;
    const int w = threadIdx.x >> 6;
    unsigned long long masks[R];
    int wave_total = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (sid[r] >= 0) hit_id[sid[r]] = best[r];
      masks[r] = __ballot(sid[r] >= 0 && best[r] >= 0);
      wave_total += __popcll(masks[r]);
    }
    if (lane == 0) wtot[w] = wave_total;
    __syncthreads();
    if (threadIdx.x == 0) {
      int tot = wtot[0] + wtot[1] + wtot[2] + wtot[3];
      gbase = tot ? atomicAdd(&cnt->q_shade, tot) : 0;
    }
    __syncthreads();
    int pos = gbase;
    for (int q = 0; q < w; ++q) pos += wtot[q];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (sid[r] >= 0 && best[r] >= 0) q_shade[pos + __popcll(masks[r] & ((1ull << lane) - 1ull))] = sid[r];
      pos += __popcll(masks[r]);
    }
  } else {
    int* tri_hit = ap->tri_hit;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      int q = base + r * 64 + lane;
      if (q < nq) tri_hit[q] = best[r];
    }
  }
}

template <int R>
__global__ void __launch_bounds__(256)
k_search_primary(const SearchArgs* __restrict__ ap, const f2* __restrict__ cone) {
  const int nq = *ap->nq_ptr;
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int base = wave * 64 * R;
  if ((int)(blockIdx.x * blockDim.x) * R >= nq) return;  // whole workgroup past the queue end
  const int npacked = *ap->npacked_ptr;
  const int nbatch = base < nq ? (npacked + TRI_BATCH - 1) / TRI_BATCH : 0;
  unsigned int n_exact = 0;

  int sid[R], best[R], upto[R];
  float dx[R], dy[R], dz[R];
  const float qnan = __builtin_nanf("");
  search_prologue<R>(ap, base, nq, lane, sid, dx, dy, dz);
#pragma unroll
  for (int r = 0; r < R; ++r) { best[r] = -1; upto[r] = 0; }

  int done_batches = nbatch;
  if (nbatch > 0) {
    f2 cur[12];
    int flag = reinterpret_cast<const int*>(cone)[CONE_FLAG];
#pragma unroll
    for (int c = 0; c < 12; ++c) cur[c] = cone[c];
    for (int b = 0; b < nbatch; ++b) {
      // request the next batch (one spare, zero-filled batch is allocated past the end)
      f2 nxt[12];
      const f2* np_ = cone + (size_t)(b + 1) * (CONE_DWORDS / 2);
#pragma unroll
      for (int c = 0; c < 12; ++c) nxt[c] = np_[c];
      int nflag = reinterpret_cast<const int*>(np_)[CONE_FLAG];

      // R independent chains per triangle pair so that consecutive instructions do not depend
      // on each other
      float mxr[R];
#pragma unroll
      for (int r = 0; r < R; ++r) mxr[r] = -3.0e38f;
#pragma unroll
      for (int pr = 0; pr < 4; ++pr) {
        f2 q[R];
#pragma unroll
        for (int r = 0; r < R; ++r) { f2 X = {dx[r], dx[r]}; q[r] = cur[pr * 3 + 0] * X; }
#pragma unroll
        for (int r = 0; r < R; ++r) { f2 Y = {dy[r], dy[r]}; q[r] = __builtin_elementwise_fma(cur[pr * 3 + 1], Y, q[r]); }
#pragma unroll
        for (int r = 0; r < R; ++r) { f2 Z = {dz[r], dz[r]}; q[r] = __builtin_elementwise_fma(cur[pr * 3 + 2], Z, q[r]); }
#pragma unroll
        for (int r = 0; r < R; ++r) mxr[r] = __builtin_fmaxf(__builtin_fmaxf(mxr[r], q[r].x), q[r].y);
      }
      float mx = mxr[0];
#pragma unroll
      for (int r = 1; r < R; ++r) mx = __builtin_fmaxf(mx, mxr[r]);

      if (__builtin_expect(__any(mx >= 1.0f) || flag != 0, 0)) {
        // ---- rare path: look closer at the eight triangles of this batch, in list order ----
        const SearchArgs* a2 = ap;
        asm volatile("" : "+s"(a2));  // read the argument block here, not ahead of the loop
        const float* edge = a2->edge;
        for (int j = 0; j < TRI_BATCH; ++j) {
          const int g = b * TRI_BATCH + j;
          if (g >= npacked) break;
          const float* wj = reinterpret_cast<const float*>(cone) + (size_t)b * CONE_DWORDS + (j >> 1) * 6 + (j & 1);
          const float wx = wj[0], wy = wj[2], wz = wj[4];
          unsigned nom = 0;  // bit r: ray r of this lane passes the cone of triangle j
#pragma unroll
          for (int r = 0; r < R; ++r) {
            float q = __builtin_fmaf(wz, dz[r], __builtin_fmaf(wy, dy[r], wx * dx[r]));
            // !(q < 1) also lets a NaN cone (wide triangle) through; finished or absent rays are excluded
            if (!(q < 1.0f) && best[r] < 0 && sid[r] >= 0) nom |= 1u << r;
          }
          if (!__any(nom != 0)) continue;
          const float* m = edge + (size_t)g * EDGE_FLOATS;
          const float m0 = m[0], m1 = m[1], m2 = m[2], m3 = m[3], m4 = m[4], m5 = m[5], m6 = m[6], m7 = m[7], m8 = m[8];
#pragma unroll
          for (int r = 0; r < R; ++r) {
            float s0 = __builtin_fmaf(m2, dz[r], __builtin_fmaf(m1, dy[r], m0 * dx[r]));
            float s1 = __builtin_fmaf(m5, dz[r], __builtin_fmaf(m4, dy[r], m3 * dx[r]));
            float s2 = __builtin_fmaf(m8, dz[r], __builtin_fmaf(m7, dy[r], m6 * dx[r]));
            float e = __builtin_fmaxf(__builtin_fmaxf(s0, s1), s2);
            if (!(e <= 1.0f)) nom &= ~(1u << r);
          }
          if (!__any(nom != 0)) continue;
          // FP64 decision, one ray slot at a time.  The loop over r is deliberately NOT unrolled:
          // one copy of the double-precision code keeps the kernel's register count (and with it
          // the occupancy of the steady-state loop) low; r is wave-uniform, so the selects below
          // are scalar branches.
          const int ti = checked(a2->packed_tri[g], a2->n_tris, a2->cnt, 2);
#pragma nounroll
          for (int r = 0; r < R; ++r) {
            const bool mine = (nom >> r) & 1u;
            if (!__any(mine)) continue;
            int s = sid[0];
#pragma unroll
            for (int rr = 1; rr < R; ++rr) if (rr == r) s = sid[rr];
            bool hit = false;
            double t = 0.0;
            if (mine) {
              const rtmi_triangle& tr = a2->tris[ti];
              const double* dp = a2->dirs + (size_t)s * 3;
              d3 P;
              n_exact += 1;
              hit = rtx::tri_hit(rtx::ld3(tr.v[0]), rtx::ld3(tr.v[1]), rtx::ld3(tr.v[2]), a2->planes[ti],
                                 rtx::ld3(a2->cam), rtx::mk(dp[0], dp[1], dp[2]), &t, &P);
              if (hit) a2->tval[s] = t;  // the hit point is origin + t*direction; shading rebuilds it
            }
#pragma unroll
            for (int rr = 0; rr < R; ++rr) {
              if (rr == r && hit) {
                best[rr] = ti;
                upto[rr] = g + 1;  // list positions this ray consumed
                dx[rr] = dy[rr] = dz[rr] = qnan;
              }
            }
          }
        }
        bool all_done = true;
#pragma unroll
        for (int r = 0; r < R; ++r) all_done = all_done && (best[r] >= 0 || sid[r] < 0);
        if (__all(all_done)) { done_batches = b + 1; break; }
      }
#pragma unroll
      for (int c = 0; c < 12; ++c) cur[c] = nxt[c];
      flag = nflag;
    }
  }

  // a ray with a hit consumed the list up to that hit; a ray without one consumed every
  // packed triangle the wave walked
  unsigned long long walked = (unsigned long long)min(done_batches * TRI_BATCH, npacked);
  unsigned long long tests = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (sid[r] >= 0) tests += best[r] >= 0 ? (unsigned long long)upto[r] : walked;
  }
  DevCounters* cnt = ap->cnt;
  wave_count(cnt, C_TESTS_PRIMARY, tests);
  wave_count(cnt, C_EXACT, (unsigned long long)n_exact);

  search_epilogue<R>(ap, base, nq, lane, sid, best);
}

// ---------------------------------------------------------------------------------------
// the default search: cone tree (rtmi_tree.h) in three launches
// ---------------------------------------------------------------------------------------
//   k_search_tree    FP32 only: walks the tree once per wave and emits every (ray, triangle slot)
//                    pair that passes the triangle's cone and edge tests -- the candidates
//   k_search_exact   one lane per candidate: the FP64 reference-order test; a hit lowers the
//                    ray's best list position (atomicMin)
//   k_search_finish  one lane per ray: best list position -> triangle index, t, shade queue
// Splitting the double-precision work off keeps the walk at a register count that allows eight
// waves per SIMD (the walk is a chain of dependent scalar loads; occupancy hides their latency)
// and runs the FP64 tests with all 64 lanes busy.
constexpr int CAND_STAGE = 512;  // candidates a wave collects in LDS before it reserves global space

// Evaluate one record of inner-node cones for the R rays of every lane; returns the mask of
// children that some ray of the wave passes, or that are flagged "always descend".
template <int R>
__device__ __forceinline__ unsigned tree_eval(const f2* __restrict__ rp, const float (&dx)[R], const float (&dy)[R],
                                              const float (&dz)[R]) {
  f2 cur[12];
#pragma unroll
  for (int c = 0; c < 12; ++c) cur[c] = rp[c];
  unsigned mask = reinterpret_cast<const unsigned*>(rp)[REC_ALWAYS];
#pragma unroll
  for (int pr = 0; pr < 4; ++pr) {
    f2 q[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { f2 X = {dx[r], dx[r]}; q[r] = cur[pr * 3 + 0] * X; }
#pragma unroll
    for (int r = 0; r < R; ++r) { f2 Y = {dy[r], dy[r]}; q[r] = __builtin_elementwise_fma(cur[pr * 3 + 1], Y, q[r]); }
#pragma unroll
    for (int r = 0; r < R; ++r) { f2 Z = {dz[r], dz[r]}; q[r] = __builtin_elementwise_fma(cur[pr * 3 + 2], Z, q[r]); }
    float ma = q[0].x, mb = q[0].y;  // fmaxf drops the NaN of absent rays
#pragma unroll
    for (int r = 1; r < R; ++r) { ma = __builtin_fmaxf(ma, q[r].x); mb = __builtin_fmaxf(mb, q[r].y); }
    if (__any(ma >= 1.0f)) mask |= 1u << (2 * pr);
    if (__any(mb >= 1.0f)) mask |= 2u << (2 * pr);
  }
  return (unsigned)__builtin_amdgcn_readfirstlane((int)mask);
}

// Called by the last kernel of a tree search launch (any one workgroup's first NSLOT threads do
// the work): add this launch's candidate counts to the frame total and clear the OTHER set of
// list counters, which the launch before this one used and the next one will use.
__device__ inline void search_lists_retire(DevCounters* cnt, int parity) {
  if (blockIdx.x == 0 && threadIdx.x < NSLOT) {
    atomicAdd(&cnt->cand_total, (unsigned long long)(unsigned int)cnt->cand_n[parity][threadIdx.x][0]);
    cnt->cand_n[parity ^ 1][threadIdx.x][0] = 0;
  }
}

// Hand the wave's staged candidates to its global list: one returning atomic per flush.
__device__ __forceinline__ void cand_flush(const SearchArgs* ap, const int2* stage, int fill, int lane, int list, int parity) {
  if (fill == 0) return;
  DevCounters* cnt = ap->cnt;
  const int list_cap = ap->cand_cap / NSLOT;
  int base = 0;
  if (lane == 0) base = atomicAdd(&cnt->cand_n[parity][list][0], fill);
  base = __builtin_amdgcn_readfirstlane(base);
  if (base + fill > list_cap || base < 0) {
    if (lane == 0) cnt->overflow = 1;  // the caller renders this frame again with the list scan
    return;
  }
  int2* cand = ap->cand + (size_t)list * list_cap + base;
  for (int i = lane; i < fill; i += 64) cand[i] = stage[i];
}

// Each wave walks the tree depth first for its R*64 rays.  The walk is wave-uniform: `lvl`, the
// record index `ri` and `masks` (one byte of children still to visit per level) live in scalar
// registers.  Record (l, r) holds the cones of nodes 8r..8r+7 of level l; the children of node
// (l, j) are record (l-1, j); level-0 nodes are triangle slots, and a level-0 record is not
// pushed but evaluated per lane: every (ray, slot) pair that passes becomes a candidate.
template <int R>
__global__ void __launch_bounds__(256, 6)
k_search_tree(const SearchArgs* __restrict__ ap, const f2* __restrict__ rec, const float* __restrict__ edge_all,
              const int* __restrict__ slot_pos, int parity) {
  __shared__ int2 stage_all[4][CAND_STAGE];
  const int nq = *ap->nq_ptr;
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int base = wave * 64 * R;
  if (base >= nq) return;  // no workgroup-wide synchronisation below: waves leave on their own
  int2* stage = stage_all[threadIdx.x >> 6];

  int sid[R];
  float dx[R], dy[R], dz[R];
  search_prologue<R>(ap, base, nq, lane, sid, dx, dy, dz);
  {
    int* best_pos = ap->best_pos;  // this launch decides these rays: no hit so far
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (sid[r] >= 0) best_pos[sid[r]] = 0x7fffffff;
  }
  unsigned n_leaf_rec = 0, n_node_rec = 0;
  int fill = 0;

  const int ltop = ap->tree.ltop, n_root = ap->tree.n_root;
  // first record of each level, in scalar registers; the select chain is scalar code
  const int o0 = ap->tree.rec_off[0], o1 = ap->tree.rec_off[1], o2 = ap->tree.rec_off[2], o3 = ap->tree.rec_off[3],
            o4 = ap->tree.rec_off[4], o5 = ap->tree.rec_off[5], o6 = ap->tree.rec_off[6], o7 = ap->tree.rec_off[7];
  static_assert(TREE_MAX_LEVELS == 8, "level_off lists eight levels");
  auto level_off = [&](int lvl) {
    return lvl == 1 ? o1 : lvl == 2 ? o2 : lvl == 3 ? o3 : lvl == 4 ? o4 : lvl == 5 ? o5 : lvl == 6 ? o6 : o7;
  };
  // Level-0 record `ri`: the triangle cones of slots 8*ri..8*ri+7 against every ray of the lane.
  // A triangle that no ray of the wave passes costs three maxima and a vote; for the others the
  // edge test follows at once, and what remains is a candidate.
  const int n_slots = ap->tree.n_slots, n_records = ap->n_records;
  auto leaf = [&](int ri) {
    if ((unsigned)(ri * 8) >= (unsigned)n_slots) {  // only a defective record can lead here
      ap->cnt->bug = 3;
      return;
    }
    ++n_leaf_rec;
    const f2* rp = rec + (size_t)(o0 + ri) * (REC_DWORDS / 2);
    f2 cur[12];
#pragma unroll
    for (int c = 0; c < 12; ++c) cur[c] = rp[c];
    const unsigned always = reinterpret_cast<const unsigned*>(rp)[REC_ALWAYS];  // wide triangles (NaN cones)
    const float* edge = edge_all + (size_t)ri * (8 * EDGE_FLOATS);
#pragma unroll
    for (int pr = 0; pr < 4; ++pr) {
      f2 q[R];
#pragma unroll
      for (int r = 0; r < R; ++r) { f2 X = {dx[r], dx[r]}; q[r] = cur[pr * 3 + 0] * X; }
#pragma unroll
      for (int r = 0; r < R; ++r) { f2 Y = {dy[r], dy[r]}; q[r] = __builtin_elementwise_fma(cur[pr * 3 + 1], Y, q[r]); }
#pragma unroll
      for (int r = 0; r < R; ++r) { f2 Z = {dz[r], dz[r]}; q[r] = __builtin_elementwise_fma(cur[pr * 3 + 2], Z, q[r]); }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int jj = pr * 2 + h;
        float mq = h ? q[0].y : q[0].x;  // fmaxf drops the NaN of absent rays
#pragma unroll
        for (int r = 1; r < R; ++r) mq = __builtin_fmaxf(mq, h ? q[r].y : q[r].x);
        if (!__any(mq >= 1.0f) && !((always >> jj) & 1u)) continue;
        const float* m = edge + jj * EDGE_FLOATS;
        const float m0 = m[0], m1 = m[1], m2 = m[2], m3 = m[3], m4 = m[4], m5 = m[5], m6 = m[6], m7 = m[7], m8 = m[8];
        const int pos = checked(slot_pos[ri * 8 + jj], ap->n_group, ap->cnt, 4);  // the triangle's position in the group's list order
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const float qv = h ? q[r].y : q[r].x;
          float s0 = __builtin_fmaf(m2, dz[r], __builtin_fmaf(m1, dy[r], m0 * dx[r]));
          float s1 = __builtin_fmaf(m5, dz[r], __builtin_fmaf(m4, dy[r], m3 * dx[r]));
          float s2 = __builtin_fmaf(m8, dz[r], __builtin_fmaf(m7, dy[r], m6 * dx[r]));
          float e = __builtin_fmaxf(__builtin_fmaxf(s0, s1), s2);
          // !(q < 1) also lets a NaN cone (wide triangle) through; absent rays are excluded by sid
          const bool pass = !(qv < 1.0f) && sid[r] >= 0 && e <= 1.0f;
          const unsigned long long bm = __ballot(pass);
          if (bm == 0) continue;
          const int n = __popcll(bm);
          if (fill + n > CAND_STAGE) {
            cand_flush(ap, stage, fill, lane, wave & (NSLOT - 1), parity);
            fill = 0;
            __builtin_amdgcn_wave_barrier();
          }
          if (pass) stage[fill + __popcll(bm & ((1ull << lane) - 1ull))] = make_int2(sid[r], pos);
          fill += n;
        }
      }
    }
  };

  if (ltop == 0) {
    for (int t = 0; t < n_root; ++t) leaf(t);
  } else {
    for (int t = 0; t < n_root; ++t) {
      int lvl = ltop, ri = t;
      unsigned long long masks;
      {
        unsigned m = tree_eval<R>(rec + (size_t)checked(level_off(lvl) + ri, n_records, ap->cnt, 5) * (REC_DWORDS / 2), dx, dy, dz);
        masks = (unsigned long long)m << (8 * lvl);
        ++n_node_rec;
      }
      for (;;) {
        const unsigned cur = (unsigned)(masks >> (8 * lvl)) & 0xffu;
        if (cur == 0) {
          if (lvl == ltop) break;
          ++lvl;
          ri >>= 3;
          continue;
        }
        const int c = __builtin_ctz(cur);
        masks &= ~(1ull << (8 * lvl + c));
        const int child = ri * 8 + c;
        if (lvl == 1) {
          leaf(child);
        } else {
          --lvl;
          ri = child;
          unsigned m = tree_eval<R>(rec + (size_t)checked(level_off(lvl) + ri, n_records, ap->cnt, 5) * (REC_DWORDS / 2), dx, dy, dz);
          masks |= (unsigned long long)m << (8 * lvl);
          ++n_node_rec;
        }
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
  cand_flush(ap, stage, fill, lane, wave & (NSLOT - 1), parity);

  // work counters: (ray, triangle) cone tests at leaf level, (ray, node) tests above it
  int active = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) active += __popcll(__ballot(sid[r] >= 0));
  DevCounters* cnt = ap->cnt;
  wave_count(cnt, C_TESTS_PRIMARY, lane == 0 ? 8ull * n_leaf_rec * (unsigned long long)active : 0ull);
  wave_count(cnt, C_NODE_TESTS, lane == 0 ? 8ull * n_node_rec * (unsigned long long)active : 0ull);
}

// One lane per candidate: the reference's own arithmetic in FP64; a hit competes for the ray's
// lowest list position.  Workgroup w serves candidate list w & 63 together with the other
// workgroups of that residue (the grid is a multiple of 64 workgroups).
__global__ void __launch_bounds__(256)
k_search_exact(const SearchArgs* __restrict__ ap, int parity) {
  const DevCounters* cnt = ap->cnt;
  unsigned int n_exact = 0;
  if (!cnt->overflow) {  // after an overflow the lists are incomplete and the frame is rendered again
    const int list = blockIdx.x & (NSLOT - 1), part = blockIdx.x / NSLOT, parts = gridDim.x / NSLOT;
    const int list_cap = ap->cand_cap / NSLOT;
    const int n = min(cnt->cand_n[parity][list][0], list_cap);
    const int2* cand = ap->cand + (size_t)list * list_cap;
    const double* dirs = ap->dirs;
    const d3 o = rtx::ld3(ap->cam);
    // The entry of the NEXT step is fetched (with the triangle index it leads to) before the
    // current one is evaluated: a step begins with a chain of dependent loads, and this way the
    // chain of step k+1 runs under the arithmetic of step k.
    const int stride = parts * blockDim.x;
    const int* group = ap->group;
    int i = part * blockDim.x + threadIdx.x;
    DevCounters* wcnt = ap->cnt;
    const int n_group = ap->n_group, n_samples = ap->n_samples, n_tris = ap->n_tris;
    auto entry = [&](int k) {  // candidate k with both indices range-checked
      int2 e = cand[k];
      e.x = checked(e.x, n_samples, wcnt, 6);
      e.y = checked(e.y, n_group, wcnt, 7);
      return e;
    };
    int2 c = i < n ? entry(i) : make_int2(0, 0);
    int ti = i < n ? checked(group[c.y], n_tris, wcnt, 8) : 0;
    while (i < n) {
      const int inext = i + stride;
      const int2 cn = inext < n ? entry(inext) : c;
      const int tin = checked(group[cn.y], n_tris, wcnt, 8);
      const int s = c.x, pos = c.y;
      const double* dp = dirs + (size_t)s * 3;
      const rtmi_triangle& tr = ap->tris[ti];
      d3 P;
      double tv;
      n_exact += 1;
      if (rtx::tri_hit(rtx::ld3(tr.v[0]), rtx::ld3(tr.v[1]), rtx::ld3(tr.v[2]), ap->planes[ti], o,
                       rtx::mk(dp[0], dp[1], dp[2]), &tv, &P))
        atomicMin(&ap->best_pos[s], pos);
      c = cn;
      ti = tin;
      i = inext;
    }
  }
  wave_count(ap->cnt, C_EXACT, (unsigned long long)n_exact);
}

// Top-level runs only (mode 1): one lane per ray of the queue, lowest list position that was
// hit -> triangle index in tri_hit[q] for k_resolve_top, and the hit's t.
__global__ void __launch_bounds__(256)
k_search_finish(const SearchArgs* __restrict__ ap, int parity) {
  const int nq = *ap->nq_ptr;
  search_lists_retire(ap->cnt, parity);
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  const int s = checked(ap->queue[q], ap->n_samples, ap->cnt, 12);
  int ti = -1;
  const int pos = ap->best_pos[s];
  // After a candidate-list overflow the frame is rendered again; all this launch must do then is
  // leave valid indices behind.  The range check keeps any unexpected value from becoming an address.
  if (!ap->cnt->overflow && (unsigned)pos < (unsigned)ap->n_group) {
    ti = checked(ap->group[pos], ap->n_tris, ap->cnt, 13);
    const double* dp = ap->dirs + (size_t)s * 3;
    const double tv = rtx::tri_t(ap->planes[ti], rtx::ld3(ap->cam), rtx::mk(dp[0], dp[1], dp[2]));
    ap->tval[s] = tv;  // the hit point is origin + t*direction; shading rebuilds it
  }
  ap->tri_hit[q] = ti;
}

// ---------------------------------------------------------------------------------------
// k_shade: calculateLighting (RayTracingFunctions.swift:182-271) for every hit sample
// ---------------------------------------------------------------------------------------
struct RGBd {
  double r, g, b;
};

// the two frame constants shading reads
struct ShadeConst {
  double cam[3];
  double ambient[3];
};

// pow(base, exponent) of the specular term (RayTracingFunctions.swift:259), base in (0, 1].
// For the usual small whole-number exponents (OBJ "Ns" values like 8) binary powering is used:
// at most ceil(log2 n)+popcount roundings, i.e. a few ulp, far below what can change
// UInt8(255 * c); every other exponent goes through the device pow().
__device__ inline double spec_pow(double base, double e) {
  if (e >= 1.0 && e <= 64.0 && e == floor(e)) {
    int n = (int)e;
    double r = 1.0, b = base;
    bool first = true;
    while (n) {
      if (n & 1) { r = first ? b : r * b; first = false; }
      n >>= 1;
      if (n) b = b * b;
    }
    return r;
  }
  return pow(base, e);
}

// Number of lights with normalLightVectorDp > 0 (RayTracingFunctions.swift:227-232) at a hit whose
// object has no material.  There the light loop adds nothing to the colour (:245), but every
// such light still costs the reference one shadow ray (:236), which the counters report.  The
// sign of dp(normalize(L), normalize(N)) equals the sign of dp(L, N) whenever |dp(L, N)| exceeds
// 2^-40 |L|_1 |N|_1 (the rounding of the normalised form is below 2^-48); only inside that band
// are the two normalisations evaluated.
__device__ inline int count_facing_lights(const SceneDev& sc, d3 P, d3 n, bool n_unit) {
  int c = 0;
  const double n1 = fabs(n.x) + fabs(n.y) + fabs(n.z);
  for (int li = 0; li < sc.n_lights; ++li) {
    d3 L = rtx::sub(rtx::ld3(sc.lights[li].loc), P);
    double sdp = rtx::dp(L, n);
    double band = 9.094947017729282e-13 * ((fabs(L.x) + fabs(L.y) + fabs(L.z)) * n1);
    if (sdp > band) { ++c; continue; }
    if (sdp < -band) continue;
    d3 nn = n_unit ? n : rtx::normalize(n);
    if (rtx::dp(rtx::normalize(L), nn) > 0) ++c;
  }
  return c;
}

template <bool REFLECT, bool HAS_TOP>
__device__ RGBd calculate_lighting(const SceneDev& sc, const TopTreeRef& tt, const ShadeConst& fc, const Hit& is, d3 camera,
                                   unsigned long long* n_refl, unsigned long long* n_shadow,
                                   unsigned long long* n_tests, unsigned long long* n_elided) {
  d3 surface_point = is.P;
  RGBd I = {0.0, 0.0, 0.0};
  const bool has_material = is.material >= 0;
  rtmi_material mat;
  if (has_material) {
    mat = sc.materials[is.material];
    // intensityMultiplier = ambientLight; intensityMultiplier.scale(material.ambient)  RTF:193-196
    I.r = mat.Ka[0] * fc.ambient[0];
    I.g = mat.Ka[1] * fc.ambient[1];
    I.b = mat.Ka[2] * fc.ambient[2];
  }
  if (!is.has_normal) return I;  // RTF:199-202
  if (!REFLECT && !has_material) {
    // nothing below can change I (= 0); only the shadow rays are counted
    int c = count_facing_lights(sc, surface_point, is.n, is.n_unit);
    *n_shadow += c;
    *n_elided += c;
    return I;
  }
  d3 normal = is.n;
  // RTF:204; only the reflection direction and the specular term read it
  d3 surface_to_camera = rtx::mk(0.0, 0.0, 0.0);
  if (REFLECT || has_material) surface_to_camera = rtx::normalize(rtx::sub(camera, surface_point));

  if constexpr (REFLECT) {  // RTF:207-224
    d3 incident = rtx::neg(surface_to_camera);
    d3 reflection = rtx::sub(incident, rtx::smul(2 * rtx::dp(incident, normal), normal));
    RGBd contribution = {0.0, 0.0, 0.0};
    Hit rh;
    bool anb;
    *n_refl += 1;
    if (trace_top_secondary<HAS_TOP>(sc, tt, surface_point, reflection, true, &rh, &anb, n_tests)) {
      contribution = calculate_lighting<false, HAS_TOP>(sc, tt, fc, rh, surface_point, n_refl, n_shadow, n_tests, n_elided);
    }
    I.r = I.r + contribution.r;
    I.g = I.g + contribution.g;
    I.b = I.b + contribution.b;
  }

  if (!has_material) {  // the light loop adds nothing (RTF:245); count its shadow rays
    int c = count_facing_lights(sc, surface_point, normal, true);
    *n_shadow += c;
    *n_elided += c;
    I.r = rtx::clamp01(I.r); I.g = rtx::clamp01(I.g); I.b = rtx::clamp01(I.b);  // RTF:269
    return I;
  }
#pragma nounroll
  for (int li = 0; li < sc.n_lights; ++li) {  // RTF:226-268
    const rtmi_light& light = sc.lights[li];
    d3 point_to_light = rtx::normalize(rtx::sub(rtx::ld3(light.loc), surface_point));
    double ndl = rtx::dp(point_to_light, normal);
    if (ndl <= 0) continue;
    Hit dummy;
    bool anb = false;
    *n_shadow += 1;
    trace_top_secondary<HAS_TOP>(sc, tt, surface_point, point_to_light, false, &dummy, &anb, n_tests);
    if (anb) continue;           // RTF:241
    if (!has_material) continue; // RTF:245
    double lr = light.diffuse[0], lg = light.diffuse[1], lb = light.diffuse[2];
    lr = ndl * lr; lg = ndl * lg; lb = ndl * lb;                    // scale(normalLightVectorDp)
    lr = mat.Kd[0] * lr; lg = mat.Kd[1] * lg; lb = mat.Kd[2] * lb;  // scale(material.diffuse)
    I.r = I.r + lr; I.g = I.g + lg; I.b = I.b + lb;
    d3 light_reflection = rtx::normalize(rtx::sub(rtx::smul(2 * ndl, normal), point_to_light));  // RTF:251
    double rv = rtx::dp(light_reflection, surface_to_camera);
    if (rv <= 0) continue;  // RTF:255-257
    double specular_term = spec_pow(rv, mat.Ns);  // RTF:259
    double sr = light.specular[0], sg = light.specular[1], sb = light.specular[2];
    sr = specular_term * sr; sg = specular_term * sg; sb = specular_term * sb;
    sr = mat.Ks[0] * sr; sg = mat.Ks[1] * sg; sb = mat.Ks[2] * sb;
    sr = rtx::clamp01(sr); sg = rtx::clamp01(sg); sb = rtx::clamp01(sb);  // RTF:263
    I.r = I.r + sr; I.g = I.g + sg; I.b = I.b + sb;
  }
  I.r = rtx::clamp01(I.r); I.g = rtx::clamp01(I.g); I.b = rtx::clamp01(I.b);  // RTF:269
  return I;
}

// Work tallies of a shading lane, summed per wave at the end of the kernel.
struct ShadeTally {
  unsigned long long n_refl = 0, n_shadow = 0, n_hit = 0, n_tests = 0, n_elided = 0;
};

__device__ inline void shade_tally_flush(DevCounters* cnt, const ShadeTally& t) {
  wave_count(cnt, C_HITS, t.n_hit);
  wave_count(cnt, C_REFL, t.n_refl);
  wave_count(cnt, C_SHADOW, t.n_shadow);
  wave_count(cnt, C_TESTS_SECONDARY, t.n_tests);
  wave_count(cnt, C_SHADOW_ELIDED, t.n_elided);
}

// Shade sample s whose primary ray (direction d) hit `id` (triangle index, or -2 - object index);
// t_tri is the triangle's intersection parameter.
template <bool HAS_TOP>
__device__ __forceinline__ void shade_sample(const ShadeConst& fc, const SceneDev& sc, const TopTreeRef& tt, int s, d3 d,
                                             int id, double t_tri, double* rgb, ShadeTally* tl) {
  d3 cam = rtx::ld3(fc.cam);
  Hit h;
  bool ok = false;
  if (id >= 0) {
    const rtmi_triangle& tr = sc.tris[id];
    const TriPlane pl = sc.planes[id];
    // the search already decided this hit with Triangle.intersections; rebuild its point
    // exactly as Renderables.swift:153 does: origin + intersectionParameter * direction
    ok = true;
    h.t = t_tri;
    h.P = rtx::add(cam, rtx::smul(h.t, d));
    h.n = rtx::mk(pl.nx, pl.ny, pl.nz);
    h.has_normal = 1;
    h.n_unit = 1;
    h.material = tr.material;
    h.id = id;
  } else {
    const rtmi_object& ob = sc.objects[-2 - id];
    if (ob.kind == RTMI_LIGHT) {
      ok = rtx::light_hit(rtx::ld3(sc.lights[ob.first].loc), cam, d, &h.t, &h.P);
      h.has_normal = 0;
      h.n_unit = 1;
      h.material = -1;
    } else {
      ok = rtx::sphere_first(rtx::ld3(ob.center), ob.radius, cam, d, &h.t, &h.P, &h.n);
      h.has_normal = 1;
      h.n_unit = 1;
      h.material = ob.material;
    }
    h.id = id;
  }
  RGBd c = {0.0, 0.0, 0.0};
  if (ok) {
    tl->n_hit += 1;
    c = calculate_lighting<true, HAS_TOP>(sc, tt, fc, h, cam, &tl->n_refl, &tl->n_shadow, &tl->n_tests, &tl->n_elided);
  }
  rgb[(size_t)s * 3 + 0] = c.r;
  rgb[(size_t)s * 3 + 1] = c.g;
  rgb[(size_t)s * 3 + 2] = c.b;
}

// HAS_TOP: the scene lists triangles at top level, so secondary rays scan them (the code of
// that scan is left out of the instantiation for scenes without such runs).
template <bool HAS_TOP>
__global__ void __launch_bounds__(256)
k_shade(ShadeConst fc, SceneDev sc, TopTreeArgs ta, const f2* __restrict__ trec, const float* __restrict__ gst2,
        const int* __restrict__ tslot_pos, const double* __restrict__ dirs,
        const double* __restrict__ tval, const int* __restrict__ q_shade,
        const int* __restrict__ nq_ptr, const int* __restrict__ hit_id, double* rgb, DevCounters* cnt) {
  const TopTreeRef tt = make_top_tree(ta, trec, gst2, tslot_pos);
  ShadeTally tl;
  const int nq = *nq_ptr;
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += gridDim.x * blockDim.x) {
    int s = checked(q_shade[q], sc.n_samples, cnt, 17);
    d3 d = rtx::mk(dirs[(size_t)s * 3 + 0], dirs[(size_t)s * 3 + 1], dirs[(size_t)s * 3 + 2]);
    int id = hit_id[s];
    // a triangle index or -2 - object index, both about to be used as such
    if (id >= 0) id = checked(id, sc.n_tris, cnt, 18);
    else id = -2 - checked(-2 - id, sc.n_objects, cnt, 19);
    shade_sample<HAS_TOP>(fc, sc, tt, s, d, id, id >= 0 ? tval[s] : 0.0, rgb, &tl);
  }
  shade_tally_flush(cnt, tl);
}

// The cone tree's last stage for the children of a bounding object: one lane per candidate
// (grid-stride over the NSLOT lists, like k_search_exact).  The candidate whose list position is
// the ray's lowest hit -- k_search_exact has settled that -- is the ray's intersection: its
// triangle index goes to hit_id and the sample is shaded here, without a detour through the
// shade queue.  t is evaluated once more (the same two operations, the same value).  Samples
// without a hit keep the bounding object's marker in hit_id; k_resolve reads that as a miss.
template <bool HAS_TOP>
__global__ void __launch_bounds__(256, HAS_TOP ? 2 : 4)
k_shade_candidates(ShadeConst fc, SceneDev sc, TopTreeArgs ta, const f2* __restrict__ trec, const float* __restrict__ gst2,
                   const int* __restrict__ tslot_pos, const SearchArgs* __restrict__ ap, double* rgb, int par) {
  DevCounters* cnt = ap->cnt;
  const TopTreeRef tt = make_top_tree(ta, trec, gst2, tslot_pos);
  ShadeTally tl;
  if (!cnt->overflow) {
    const int list = blockIdx.x & (NSLOT - 1), part = blockIdx.x / NSLOT, parts = gridDim.x / NSLOT;
    const int list_cap = ap->cand_cap / NSLOT;
    const int n = min(cnt->cand_n[par][list][0], list_cap);
    const int2* cand = ap->cand + (size_t)list * list_cap;
    const double* dirs = ap->dirs;
    // as in k_search_exact: the next step's entry and what hangs on it are fetched one step ahead
    const int stride = parts * blockDim.x;
    const int* group = ap->group;
    const int* best_pos = ap->best_pos;
    int i = part * blockDim.x + threadIdx.x;
    const int n_group = ap->n_group, n_samples = ap->n_samples, n_tris = ap->n_tris;
    auto entry = [&](int k) {  // candidate k with both indices range-checked
      int2 e = cand[k];
      e.x = checked(e.x, n_samples, cnt, 9);
      e.y = checked(e.y, n_group, cnt, 10);
      return e;
    };
    int2 c = i < n ? entry(i) : make_int2(0, 0);
    int ti = i < n ? checked(group[c.y], n_tris, cnt, 11) : 0;
    int bp = i < n ? best_pos[c.x] : 0;
    while (i < n) {
      const int inext = i + stride;
      const int2 cn = inext < n ? entry(inext) : c;
      const int tin = checked(group[cn.y], n_tris, cnt, 11);
      const int bpn = best_pos[cn.x];
      const int s = c.x;
      if (bp == c.y) {  // else another triangle of this ray is listed earlier, or the pair was no hit
        const double* dp = dirs + (size_t)s * 3;
        const d3 d = rtx::mk(dp[0], dp[1], dp[2]);
        const double tv = rtx::tri_t(sc.planes[ti], rtx::ld3(fc.cam), d);  // k_search_exact accepted this pair
        ap->hit_id[s] = ti;
        shade_sample<HAS_TOP>(fc, sc, tt, s, d, ti, tv, rgb, &tl);
      }
      c = cn;
      ti = tin;
      bp = bpn;
      i = inext;
    }
  }
  search_lists_retire(cnt, par);
  shade_tally_flush(cnt, tl);
}

// ---------------------------------------------------------------------------------------
// k_resolve: per pixel average and RGBA8 store (RayTracingFunctions.swift:105-111,
// Utilities.swift:54-65)
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_resolve(FrameConst fc, const int2* __restrict__ blocks, int npix, const int* __restrict__ hit_id,
          const double* __restrict__ rgb, const rtmi_object* __restrict__ objects, int n_objects, DevCounters* cnt,
          unsigned char* rgba, int* hit_out) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= npix) return;
  int b = p >> 6, pix = p & (BLK_PIX - 1);
  int x, y;
  if (!pixel_coords(fc, blocks, b, pix, &x, &y)) return;
  double cr = 0.0, cg = 0.0, cb = 0.0;  // cumulative = RGB(0,0,0), Utilities.swift:56
  for (int q = 0; q < fc.spp; ++q) {
    int s = ((b * fc.spp + q) << 6) + pix;
    int id = hit_id[s];
    // a bounding object's marker that no child hit has replaced: the ray hit nothing (RTF:86-93)
    if (id <= -2 && objects[checked(-2 - id, n_objects, cnt, 20)].kind == RTMI_BOUNDING_SPHERE) id = -1;
    double r = 0.0, g = 0.0, bl = 0.0;  // RGB.black for a miss
    if (id != -1) {
      r = rgb[(size_t)s * 3 + 0];
      g = rgb[(size_t)s * 3 + 1];
      bl = rgb[(size_t)s * 3 + 2];
    }
    cr = cr + r; cg = cg + g; cb = cb + bl;
    if (hit_out) hit_out[((size_t)y * fc.W + x) * fc.spp + q] = id;
  }
  double factor = 1 / (double)fc.spp;  // Utilities.swift:61-62
  double ar = factor * cr, ag = factor * cg, ab = factor * cb;
  size_t first_byte = (size_t)4 * fc.W * y + (size_t)x * 4;  // RayTracingFunctions.swift:106-107
  unsigned int px = (unsigned int)(unsigned char)(255 * ar) | ((unsigned int)(unsigned char)(255 * ag) << 8) |
                    ((unsigned int)(unsigned char)(255 * ab) << 16) | (255u << 24);
  *reinterpret_cast<unsigned int*>(rgba + first_byte) = px;  // R,G,B,255 in memory order (:108-111)
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
struct TriGroup {
  int object;          // owning object index (bounding sphere), or -1 for the top-level runs
  int n = 0;           // triangles in the group
  int* d_group = nullptr;       // ordinal -> triangle index
  unsigned char* d_keep = nullptr;
  int* d_block_counts = nullptr;
  float* d_cone = nullptr;      // (n/8 + 2) batches of CONE_DWORDS
  float* d_edge = nullptr;      // n x EDGE_FLOATS
  int* d_packed_tri = nullptr;
  int* d_npacked = nullptr;     // triangles kept by the last pack
  size_t cone_bytes = 0;
  // cone tree (rtmi_tree.h): spatial slot order fixed at scene creation, cones rebuilt per frame
  TreeDev tree{};
  int* d_slot_tri = nullptr;
  int* d_slot_pos = nullptr;
  float* d_rec = nullptr;
  float* d_tedge = nullptr;
  NodeCone* d_nodes = nullptr;
  unsigned int* d_ticket = nullptr;
  size_t n_records = 0;  // records of the tree, all levels
  TreeHost tree_host;    // host copy of the slot order, kept until the scene is built
  // which operands the device holds: 0 none, 1 cone tree, 2 list scan -- and for which camera position
  int packed_for = 0;
  double packed_cam[3] = {0, 0, 0};
};

struct rtmi_scene {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev_a = nullptr, ev_b = nullptr, ev_c = nullptr, ev_d = nullptr;
  std::vector<rtmi_object> objects;
  int n_lights = 0, n_materials = 0;
  long long n_tris = 0;
  rtmi_object* d_objects = nullptr;
  rtmi_light* d_lights = nullptr;
  rtmi_material* d_materials = nullptr;
  rtmi_triangle* d_tris = nullptr;
  TriPlane* d_planes = nullptr;
  int* d_tri_owner = nullptr;
  float* d_trec = nullptr;      // bounding-sphere records of the top-level runs (secondary rays)
  float* d_gst2 = nullptr;      // their per-triangle plane and edge operands
  TopTreeArgs top_args{};
  std::vector<TriGroup> groups;  // [0] = top-level runs (may be empty), then bounding objects
  SceneDev dev{};
  // per-batch work buffers, grown on demand
  size_t cap_samples = 0;
  int* d_hit_id = nullptr;
  int* d_tri_hit = nullptr;
  int* d_q_child = nullptr;
  int* d_q_shade = nullptr;
  int* d_q_top = nullptr;
  double* d_rgb = nullptr;
  double* d_dirs = nullptr;
  double* d_tval = nullptr;
  int* d_best_pos = nullptr;   // tree search: per sample, lowest list position hit
  int2* d_cand = nullptr;      // tree search: candidate pairs, cap_cand entries in NSLOT lists
  size_t cap_cand = 0;
  std::vector<SearchArgs> h_sargs;   // one argument block per triangle group
  SearchArgs* d_sargs = nullptr;
  double* d_tabs = nullptr;  // utab then vtab
  size_t cap_tabs = 0;
  std::array<double, 18> tabs_key{};  // what the tables were computed for
  bool tabs_valid = false;
  std::vector<SearchArgs> sargs_uploaded;  // what d_sargs holds
  float4* d_genf = nullptr;                // k_gen pre-test operands, one per sphere-like object
  std::vector<float4> genf_uploaded;
  std::vector<rtmi_light> h_lights;
  int2* d_blocks = nullptr;
  size_t cap_blocks = 0;
  std::vector<int2> h_blocks;          // host copy of the block list last uploaded
  std::array<int, 9> blocks_key{};     // frame geometry + tile selection it was built for
  bool blocks_valid = false;
  DevCounters* d_cnt = nullptr;
  int n_plain_shadeable = 0;  // spheres and lights in the object list (hits on them go through the shade queue)
  int exact_grid = 2048;  // workgroups of the grid-stride k_search_exact: eight per compute unit
  // one frame may be in flight between rtmi_render_begin and rtmi_render_end
  bool pending = false;
  DevCounters* h_cnt = nullptr;  // pinned host copy of the device counters
  std::vector<std::array<hipEvent_t, 3>> ev_pool;  // per search launch: before, after the search, after its shading; reused
  size_t ev_used = 0;
  long long pend_launches = 0;
  int64_t pend_live_samples = 0;
  bool pend_shade_timed = false;
  rtmi_camera pend_cam{};
  rtmi_params pend_params{};
  void* pend_rgba = nullptr;
  void* pend_hit = nullptr;
  std::mutex mu;
};

static int32_t validate_scene(const rtmi_scene_desc* s) {
  if (!s) return RTMI_ERR_INVALID_ARGUMENT;
  if (s->n_lights < 0 || s->n_materials < 0 || s->n_objects < 0 || s->n_triangles < 0) return RTMI_ERR_INVALID_ARGUMENT;
  if ((s->n_lights && !s->lights) || (s->n_materials && !s->materials) || (s->n_objects && !s->objects) ||
      (s->n_triangles && !s->triangles))
    return RTMI_ERR_INVALID_ARGUMENT;
  if (s->n_triangles > 0x7ffffff0LL || s->n_objects > 0x3fffffffLL) return RTMI_ERR_UNSUPPORTED;
  for (int64_t k = 0; k < s->n_objects; ++k) {
    const rtmi_object& ob = s->objects[k];
    if (ob.material >= s->n_materials) return RTMI_ERR_INVALID_ARGUMENT;
    switch (ob.kind) {
      case RTMI_SPHERE: break;
      case RTMI_TRIANGLES:
      case RTMI_BOUNDING_SPHERE:
        if (ob.first < 0 || ob.count < 0 || ob.first + ob.count > s->n_triangles) return RTMI_ERR_INVALID_ARGUMENT;
        break;
      case RTMI_LIGHT:
        if (ob.first < 0 || ob.first >= s->n_lights) return RTMI_ERR_INVALID_ARGUMENT;
        break;
      default: return RTMI_ERR_INVALID_ARGUMENT;
    }
  }
  for (int64_t i = 0; i < s->n_triangles; ++i)
    if (s->triangles[i].material >= s->n_materials) return RTMI_ERR_INVALID_ARGUMENT;
  return RTMI_OK;
}

static int32_t alloc_group(TriGroup& g, const std::vector<int>& idx, const rtmi_triangle* tris) {
  // An empty group (a bounding sphere without children) still gets its buffers: the search
  // kernel then runs with npacked == 0 and resolves every queued sample as a miss.
  g.n = (int)idx.size();
  const size_t n1 = std::max<size_t>(g.n, 1);
  const int nblocks = (int)((n1 + 255) / 256);
  HIP_TRY(hipMalloc(&g.d_group, sizeof(int) * n1));
  if (g.n) HIP_TRY(hipMemcpy(g.d_group, idx.data(), sizeof(int) * g.n, hipMemcpyHostToDevice));
  HIP_TRY(hipMalloc(&g.d_keep, n1));
  HIP_TRY(hipMalloc(&g.d_block_counts, sizeof(int) * nblocks));
  g.cone_bytes = sizeof(float) * CONE_DWORDS * ((size_t)(g.n + TRI_BATCH - 1) / TRI_BATCH + 2);
  HIP_TRY(hipMalloc(&g.d_cone, g.cone_bytes));
  HIP_TRY(hipMemset(g.d_cone, 0, g.cone_bytes));
  HIP_TRY(hipMalloc(&g.d_edge, sizeof(float) * EDGE_FLOATS * n1));
  HIP_TRY(hipMalloc(&g.d_packed_tri, sizeof(int) * n1));
  HIP_TRY(hipMalloc(&g.d_npacked, sizeof(int)));
  HIP_TRY(hipMemset(g.d_npacked, 0, sizeof(int)));

  TreeHost& th = g.tree_host;
  tree_build_host(tris, idx, &th);
  g.tree = th.dev;
  g.n_records = th.n_records;
  const size_t ns = th.slot_tri.size();
  HIP_TRY(hipMalloc(&g.d_slot_tri, sizeof(int) * ns));
  HIP_TRY(hipMemcpy(g.d_slot_tri, th.slot_tri.data(), sizeof(int) * ns, hipMemcpyHostToDevice));
  HIP_TRY(hipMalloc(&g.d_slot_pos, sizeof(int) * ns));
  HIP_TRY(hipMemcpy(g.d_slot_pos, th.slot_pos.data(), sizeof(int) * ns, hipMemcpyHostToDevice));
  // entries of nodes that do not exist (the tail of a level's last record) stay zero for ever
  HIP_TRY(hipMalloc(&g.d_rec, sizeof(float) * REC_DWORDS * th.n_records));
  HIP_TRY(hipMemset(g.d_rec, 0, sizeof(float) * REC_DWORDS * th.n_records));
  HIP_TRY(hipMalloc(&g.d_tedge, sizeof(float) * EDGE_FLOATS * ns));
  HIP_TRY(hipMalloc(&g.d_nodes, sizeof(NodeCone) * th.n_node_buf));
  HIP_TRY(hipMalloc(&g.d_ticket, sizeof(unsigned int)));
  HIP_TRY(hipMemset(g.d_ticket, 0, sizeof(unsigned int)));
  g.tree.slot_tri = g.d_slot_tri;
  g.tree.slot_pos = g.d_slot_pos;
  g.tree.rec = g.d_rec;
  g.tree.edge = g.d_tedge;
  g.tree.nodes = g.d_nodes;
  g.tree.ticket = g.d_ticket;
  return RTMI_OK;
}

extern "C" int32_t rtmi_abi_version(void) { return RTMI_ABI_VERSION; }

extern "C" const char* rtmi_status_string(int32_t status) {
  switch (status) {
    case RTMI_OK: return "ok";
    case RTMI_ERR_INVALID_ARGUMENT: return "invalid argument";
    case RTMI_ERR_UNSUPPORTED: return "unsupported input";
    case RTMI_ERR_NO_DEVICE: return "no usable gfx950 device";
    case RTMI_ERR_HIP: return "HIP runtime error";
    case RTMI_ERR_OUT_OF_MEMORY: return "out of device memory";
    default: return "unknown status";
  }
}

extern "C" const char* rtmi_last_error(void) { return g_last_error.c_str(); }

extern "C" int32_t rtmi_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    g_last_error = std::string("hipGetDeviceCount: ") + hipGetErrorString(e);
    return RTMI_ERR_NO_DEVICE;
  }
  return n;
}

extern "C" int32_t rtmi_samples_per_pixel(double s) {
  if (!(s > 0.0) || s > 1.0) return RTMI_ERR_INVALID_ARGUMENT;
  int n = 0;
  while (fma((double)n, s, 0.0) < 1.0) {
    if (++n > 64) return RTMI_ERR_UNSUPPORTED;
  }
  return n * n;
}

// Sphere.init(boundingObjects:), Renderables.swift:215-246 with Triangle.getBounds :169-195.
extern "C" int32_t rtmi_bounding_sphere(const rtmi_triangle* tris, int64_t n, double center[3], double* radius) {
  if ((!tris && n > 0) || n < 0 || !center || !radius) return RTMI_ERR_INVALID_ARGUMENT;
  double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int64_t i = 0; i < n; ++i)
    for (int k = 0; k < 3; ++k)
      for (int a = 0; a < 3; ++a) {
        double c = tris[i].v[k][a];
        if (lo[a] > c) lo[a] = c;
        if (hi[a] < c) hi[a] = c;
      }
  double ext[3];
  for (int a = 0; a < 3; ++a) ext[a] = fabs(hi[a] - lo[a]);
  *radius = std::max(ext[0], std::max(ext[1], ext[2])) / 2;
  for (int a = 0; a < 3; ++a) center[a] = hi[a] - (ext[a] / 2);
  return RTMI_OK;
}

// The spatial slot order the cone tree gives a triangle list (host arithmetic only).
extern "C" int32_t rtmi_tree_slot_order(const rtmi_triangle* tris, int64_t n, int32_t* slot_pos, int64_t capacity,
                                        int64_t* n_slots) {
  if ((!tris && n > 0) || n < 0 || n > 0x7ffffff0LL || !n_slots) return RTMI_ERR_INVALID_ARGUMENT;
  std::vector<int> idx((size_t)n);
  for (int64_t i = 0; i < n; ++i) idx[(size_t)i] = (int)i;
  TreeHost th;
  tree_build_host(tris, idx, &th);
  *n_slots = (int64_t)th.slot_pos.size();
  if (slot_pos) {
    if (capacity < *n_slots) return RTMI_ERR_INVALID_ARGUMENT;
    std::copy(th.slot_pos.begin(), th.slot_pos.end(), slot_pos);
  }
  return RTMI_OK;
}

extern "C" void rtmi_scene_destroy(rtmi_scene* sc) {
  if (!sc) return;
  (void)hipSetDevice(sc->device);
  if (sc->stream) (void)hipStreamSynchronize(sc->stream);
  for (auto& g : sc->groups) {
    (void)hipFree(g.d_group); (void)hipFree(g.d_keep); (void)hipFree(g.d_block_counts);
    (void)hipFree(g.d_cone); (void)hipFree(g.d_edge); (void)hipFree(g.d_packed_tri); (void)hipFree(g.d_npacked);
    (void)hipFree(g.d_slot_tri); (void)hipFree(g.d_slot_pos); (void)hipFree(g.d_rec); (void)hipFree(g.d_tedge);
    (void)hipFree(g.d_nodes); (void)hipFree(g.d_ticket);
  }
  (void)hipFree(sc->d_objects); (void)hipFree(sc->d_lights); (void)hipFree(sc->d_materials);
  (void)hipFree(sc->d_tris); (void)hipFree(sc->d_planes); (void)hipFree(sc->d_tri_owner);
  (void)hipFree(sc->d_trec); (void)hipFree(sc->d_gst2);
  (void)hipFree(sc->d_hit_id); (void)hipFree(sc->d_tri_hit); (void)hipFree(sc->d_q_child);
  (void)hipFree(sc->d_q_shade); (void)hipFree(sc->d_q_top); (void)hipFree(sc->d_rgb);
  (void)hipFree(sc->d_dirs); (void)hipFree(sc->d_tval); (void)hipFree(sc->d_best_pos); (void)hipFree(sc->d_cand); (void)hipFree(sc->d_tabs); (void)hipFree(sc->d_sargs); (void)hipFree(sc->d_genf);
  (void)hipFree(sc->d_blocks); (void)hipFree(sc->d_cnt);
  if (sc->h_cnt) (void)hipHostFree(sc->h_cnt);
  for (auto& tr : sc->ev_pool) for (hipEvent_t e : tr) (void)hipEventDestroy(e);
  if (sc->ev_a) (void)hipEventDestroy(sc->ev_a);
  if (sc->ev_b) (void)hipEventDestroy(sc->ev_b);
  if (sc->ev_c) (void)hipEventDestroy(sc->ev_c);
  if (sc->ev_d) (void)hipEventDestroy(sc->ev_d);
  if (sc->stream) (void)hipStreamDestroy(sc->stream);
  delete sc;
}

static int32_t scene_create_impl(const rtmi_scene_desc* desc, int32_t device, rtmi_scene* sc) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    g_last_error = "no HIP device visible";
    return RTMI_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= ndev) return RTMI_ERR_INVALID_ARGUMENT;
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) {
    g_last_error = std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only";
    return RTMI_ERR_NO_DEVICE;
  }
  sc->device = device;
  {
    // workgroups of the grid-stride kernels: RTMI_GRID_PER_CU per compute unit (default 8), a multiple of NSLOT
    const char* e = getenv("RTMI_GRID_PER_CU");
    int per_cu = e ? atoi(e) : 8;
    if (per_cu < 1 || per_cu > 64) per_cu = 8;
    sc->exact_grid = ((std::max(1, prop.multiProcessorCount) * per_cu + NSLOT - 1) / NSLOT) * NSLOT;
  }
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hipStreamCreateWithFlags(&sc->stream, hipStreamNonBlocking));
  HIP_TRY(hipEventCreate(&sc->ev_a));
  HIP_TRY(hipEventCreate(&sc->ev_b));
  HIP_TRY(hipEventCreate(&sc->ev_c));
  HIP_TRY(hipEventCreate(&sc->ev_d));

  sc->objects.assign(desc->objects, desc->objects + desc->n_objects);
  sc->h_lights.assign(desc->lights, desc->lights + desc->n_lights);
  sc->n_lights = desc->n_lights;
  sc->n_materials = desc->n_materials;
  sc->n_tris = desc->n_triangles;

  auto upload = [&](void** dst, const void* src, size_t bytes) -> int32_t {
    *dst = nullptr;
    if (bytes == 0) return RTMI_OK;
    HIP_TRY(hipMalloc(dst, bytes));
    HIP_TRY(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
    return RTMI_OK;
  };
  int32_t rc;
  if ((rc = upload((void**)&sc->d_objects, desc->objects, sizeof(rtmi_object) * desc->n_objects))) return rc;
  if ((rc = upload((void**)&sc->d_lights, desc->lights, sizeof(rtmi_light) * desc->n_lights))) return rc;
  if ((rc = upload((void**)&sc->d_materials, desc->materials, sizeof(rtmi_material) * desc->n_materials))) return rc;
  if ((rc = upload((void**)&sc->d_tris, desc->triangles, sizeof(rtmi_triangle) * desc->n_triangles))) return rc;

  // triangle -> owning object; groups
  std::vector<int> owner((size_t)desc->n_triangles, 0x7fffffff);
  std::vector<int> top;
  sc->groups.clear();
  sc->groups.emplace_back();
  sc->groups[0].object = -1;
  int n_sphere_like = 0;
  std::vector<std::vector<int>> child_lists;
  for (int64_t k = 0; k < desc->n_objects; ++k) {
    const rtmi_object& ob = desc->objects[k];
    if (ob.kind == RTMI_TRIANGLES) {
      for (int64_t i = ob.first; i < ob.first + ob.count; ++i) {
        top.push_back((int)i);
        // a triangle listed by several runs keeps its first (lowest) position
        owner[i] = std::min(owner[i], (int)k);
      }
    } else {
      ++n_sphere_like;
      if (ob.kind != RTMI_BOUNDING_SPHERE) ++sc->n_plain_shadeable;
      if (ob.kind == RTMI_BOUNDING_SPHERE) {
        TriGroup g;
        g.object = (int)k;
        sc->groups.push_back(g);
        std::vector<int> idx;
        for (int64_t i = ob.first; i < ob.first + ob.count; ++i) idx.push_back((int)i);
        child_lists.push_back(std::move(idx));
      }
    }
  }
  // a top-level triangle that appears in two runs would need per-position owners
  {
    std::vector<int> sorted = top;
    std::sort(sorted.begin(), sorted.end());
    if (std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end()) return RTMI_ERR_UNSUPPORTED;
  }
  if ((rc = alloc_group(sc->groups[0], top, desc->triangles))) return rc;
  for (size_t g = 1; g < sc->groups.size(); ++g)
    if ((rc = alloc_group(sc->groups[g], child_lists[g - 1], desc->triangles))) return rc;
  if ((rc = upload((void**)&sc->d_tri_owner, owner.data(), sizeof(int) * owner.size()))) return rc;

  if (desc->n_triangles > 0) {
    HIP_TRY(hipMalloc(&sc->d_planes, sizeof(TriPlane) * desc->n_triangles));
    int nb = (int)((desc->n_triangles + 255) / 256);
    hipLaunchKernelGGL(k_tri_planes, dim3(nb), dim3(256), 0, sc->stream, sc->d_tris, (long long)desc->n_triangles, sc->d_planes);
    HIP_TRY(hipGetLastError());
  }
  // general-origin filter for secondary rays against the top-level triangle runs
  double c0[3] = {0.0, 0.0, 0.0};
  if (!top.empty()) {
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    double slo[3] = {INFINITY, INFINITY, INFINITY}, shi[3] = {-INFINITY, -INFINITY, -INFINITY};
    auto grow = [](double* l, double* h, const double* p, double r) {
      // non-finite coordinates (such triangles are never hit) must not reach the FP32 frame of reference
      if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2]) || !std::isfinite(r)) return;
      for (int a = 0; a < 3; ++a) { l[a] = std::min(l[a], p[a] - r); h[a] = std::max(h[a], p[a] + r); }
    };
    for (int ti : top) for (int k = 0; k < 3; ++k) grow(lo, hi, desc->triangles[ti].v[k], 0.0);
    // every surface a ray can start from: all triangles, spheres, lights listed as objects
    for (int64_t i = 0; i < desc->n_triangles; ++i) for (int k = 0; k < 3; ++k) grow(slo, shi, desc->triangles[i].v[k], 0.0);
    for (int64_t k = 0; k < desc->n_objects; ++k) {
      const rtmi_object& ob = desc->objects[k];
      if (ob.kind == RTMI_SPHERE || ob.kind == RTMI_BOUNDING_SPHERE) grow(slo, shi, ob.center, fabs(ob.radius));
      else if (ob.kind == RTMI_LIGHT) grow(slo, shi, desc->lights[ob.first].loc, 0.0);
    }
    double r_o1 = 0.0;
    for (int a = 0; a < 3; ++a) {
      c0[a] = 0.5 * (lo[a] + hi[a]);
      r_o1 += std::max(fabs(slo[a] - c0[a]), fabs(shi[a] - c0[a]));
    }
    r_o1 *= 1.01;
    if (!std::isfinite(r_o1)) {  // no finite point at all: the filters then degrade to "always nominate" via their own checks
      r_o1 = 0.0;
      c0[0] = c0[1] = c0[2] = 0.0;
    }
    const size_t nt = top.size();
    HIP_TRY(hipMalloc(&sc->d_gst2, sizeof(float) * GST2_FLOATS * nt));
    hipLaunchKernelGGL(k_pack_general, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, sc->stream, sc->d_tris,
                       sc->groups[0].d_group, (int)nt, rtx::mk(c0[0], c0[1], c0[2]), r_o1, sc->d_gst2);
    HIP_TRY(hipGetLastError());
    // bounding-sphere tree over the slot order of group 0
    std::vector<float> trec;
    tree_sphere_records(desc->triangles, sc->groups[0].tree_host, c0, r_o1, &trec);
    if ((rc = upload((void**)&sc->d_trec, trec.data(), sizeof(float) * trec.size()))) return rc;
    const TreeDev& td = sc->groups[0].tree;
    sc->top_args.ltop = td.ltop;
    sc->top_args.n_root = td.n_root;
    sc->top_args.n_slots = td.n_slots;
    sc->top_args.n_records = (int)sc->groups[0].n_records;
    for (int l = 0; l < TREE_MAX_LEVELS; ++l) sc->top_args.off[l] = td.rec_off[l];
  }
  for (auto& g : sc->groups) g.tree_host = TreeHost();  // host copies are not needed any more
  HIP_TRY(hipMalloc(&sc->d_cnt, sizeof(DevCounters)));
  HIP_TRY(hipMalloc(&sc->d_genf, sizeof(float4) * std::max(1, n_sphere_like)));
  HIP_TRY(hipHostMalloc((void**)&sc->h_cnt, sizeof(DevCounters), hipHostMallocDefault));
  sc->h_sargs.resize(sc->groups.size());
  HIP_TRY(hipMalloc(&sc->d_sargs, sizeof(SearchArgs) * sc->groups.size()));
  HIP_TRY(hipStreamSynchronize(sc->stream));

  sc->dev.objects = sc->d_objects;
  sc->dev.lights = sc->d_lights;
  sc->dev.materials = sc->d_materials;
  sc->dev.tris = sc->d_tris;
  sc->dev.planes = sc->d_planes;
  sc->dev.n_objects = (int)desc->n_objects;
  sc->dev.n_tris = (int)desc->n_triangles;
  sc->dev.n_lights = desc->n_lights;
  sc->dev.n_sphere_like = n_sphere_like;
  sc->dev.n_top_tris = (long long)top.size();
  sc->dev.top_group = sc->groups[0].d_group;
  sc->dev.tri_owner = sc->d_tri_owner;
  for (int a = 0; a < 3; ++a) sc->dev.c0[a] = c0[a];
  return RTMI_OK;
}

extern "C" int32_t rtmi_scene_create(const rtmi_scene_desc* desc, int32_t device, rtmi_scene** out) {
  if (!out) return RTMI_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  int32_t rc = validate_scene(desc);
  if (rc != RTMI_OK) return rc;
  rtmi_scene* sc = new (std::nothrow) rtmi_scene();
  if (!sc) return RTMI_ERR_OUT_OF_MEMORY;
  rc = scene_create_impl(desc, device, sc);
  if (rc != RTMI_OK) {
    rtmi_scene_destroy(sc);
    return rc;
  }
  *out = sc;
  return RTMI_OK;
}

static int32_t ensure_buffers(rtmi_scene* sc, size_t samples, size_t blocks) {
  if (samples > sc->cap_samples) {
    (void)hipFree(sc->d_hit_id); (void)hipFree(sc->d_tri_hit); (void)hipFree(sc->d_q_child);
    (void)hipFree(sc->d_q_shade); (void)hipFree(sc->d_q_top); (void)hipFree(sc->d_rgb);
    (void)hipFree(sc->d_dirs); (void)hipFree(sc->d_tval); (void)hipFree(sc->d_best_pos); (void)hipFree(sc->d_cand);
    sc->d_hit_id = sc->d_tri_hit = sc->d_q_child = sc->d_q_shade = sc->d_q_top = sc->d_best_pos = nullptr;
    sc->d_rgb = sc->d_dirs = sc->d_tval = nullptr;
    sc->d_cand = nullptr;
    sc->cap_samples = 0;
    HIP_TRY(hipMalloc(&sc->d_hit_id, sizeof(int) * samples));
    HIP_TRY(hipMalloc(&sc->d_tri_hit, sizeof(int) * samples));
    HIP_TRY(hipMalloc(&sc->d_q_child, sizeof(int) * samples));
    HIP_TRY(hipMalloc(&sc->d_q_shade, sizeof(int) * samples));
    HIP_TRY(hipMalloc(&sc->d_q_top, sizeof(int) * samples));
    HIP_TRY(hipMalloc(&sc->d_rgb, sizeof(double) * 3 * samples));
    HIP_TRY(hipMalloc(&sc->d_dirs, sizeof(double) * 3 * samples));
    HIP_TRY(hipMalloc(&sc->d_tval, sizeof(double) * samples));
    HIP_TRY(hipMalloc(&sc->d_best_pos, sizeof(int) * samples));
    // A closed mesh that fills the frame yields about one candidate per sample; twice that, plus a
    // floor that keeps each of the NSLOT lists roomy when few waves share few lists (small frames).
    sc->cap_cand = 2 * samples + (size_t)NSLOT * 16384;
    // RTMI_CAND_CAPACITY (entries, for tests): a small value makes the lists overflow, which sends the frame
    // through the list scan instead (render_end)
    if (const char* e = getenv("RTMI_CAND_CAPACITY")) {
      const long long v = atoll(e);
      if (v > 0) sc->cap_cand = (size_t)std::max<long long>(v, NSLOT);
    }
    HIP_TRY(hipMalloc(&sc->d_cand, sizeof(int2) * sc->cap_cand));
    sc->cap_samples = samples;
  }
  if (blocks > sc->cap_blocks) {
    (void)hipFree(sc->d_blocks);
    sc->d_blocks = nullptr;
    sc->cap_blocks = 0;
    sc->blocks_valid = false;
    HIP_TRY(hipMalloc(&sc->d_blocks, sizeof(int2) * blocks));
    sc->cap_blocks = blocks;
  }
  return RTMI_OK;
}

// Per-frame cone tree of one group for the frame's camera position.
static int32_t pack_group_tree(rtmi_scene* sc, TriGroup& g, d3 cam) {
  if (g.n == 0) return RTMI_OK;  // the records were zeroed at creation: nothing passes
  int nblocks = (g.tree.n_slots + TREE_BLOCK - 1) / TREE_BLOCK;
  hipLaunchKernelGGL(k_pack_tree, dim3(nblocks), dim3(TREE_BLOCK), 0, sc->stream, sc->d_tris, sc->d_planes, g.tree, cam);
  HIP_TRY(hipGetLastError());
  return RTMI_OK;
}

// Run the list-scan pack for one group against the frame's camera.
static int32_t pack_group(rtmi_scene* sc, TriGroup& g, d3 cam) {
  if (g.n == 0) return RTMI_OK;
  int nblocks = (g.n + 255) / 256;
  HIP_TRY(hipMemsetAsync(g.d_cone, 0, g.cone_bytes, sc->stream));  // zero cone vector: never nominates; flags clear
  hipLaunchKernelGGL(k_pack_flags, dim3(nblocks), dim3(256), 0, sc->stream, sc->d_tris, sc->d_planes, g.d_group, g.n,
                     cam, g.d_keep, g.d_block_counts);
  hipLaunchKernelGGL(k_pack_scan, dim3(1), dim3(1024), 0, sc->stream, g.d_block_counts, nblocks, g.d_npacked);
  hipLaunchKernelGGL(k_pack_write, dim3(nblocks), dim3(256), 0, sc->stream, sc->d_tris, g.d_group, g.n, cam, g.d_keep,
                     g.d_block_counts, g.d_cone, g.d_edge, g.d_packed_tri);
  HIP_TRY(hipGetLastError());
  return RTMI_OK;
}

static int search_rays_per_lane() {
  // rays per lane of k_search_primary; RTMI_SEARCH_R=4|8 overrides the default for experiments
  static int r = [] {
    const char* e = getenv("RTMI_SEARCH_R");
    int v = e ? atoi(e) : 4;
    return (v == 8 || v == 2) ? v : 4;
  }();
  return r;
}
constexpr int MAX_FRAME_EDGE = 1 << 20;          // widest and tallest frame accepted, in pixels
constexpr size_t MAX_BATCH_SAMPLES = 1u << 28;  // 256 Mi samples per batch (about 100 B of work buffers per sample)

// Frames with more samples than this are rendered in several batches of whole pixel blocks.
// RTMI_MAX_BATCH_SAMPLES lowers it (tests use that to take the multi-batch path with small frames).
static size_t max_batch_samples() {
  static size_t v = [] {
    const char* e = getenv("RTMI_MAX_BATCH_SAMPLES");
    long long x = e ? atoll(e) : 0;
    return x > 0 ? std::min<size_t>((size_t)x, MAX_BATCH_SAMPLES) : MAX_BATCH_SAMPLES;
  }();
  return v;
}

// Enqueue a whole frame on the scene's stream; nothing here waits for the device.  The caller
// holds sc->mu.
static int32_t render_enqueue(rtmi_scene* sc, const rtmi_camera* cam, const rtmi_params* p, void* rgba_dev,
                              void* hit_dev) {
  if (!sc || !cam || !p || !rgba_dev) return RTMI_ERR_INVALID_ARGUMENT;
  if (p->width < 2 || p->height < 2) return RTMI_ERR_INVALID_ARGUMENT;
  if (p->width > MAX_FRAME_EDGE || p->height > MAX_FRAME_EDGE) return RTMI_ERR_UNSUPPORTED;  // index arithmetic below is sized for this
  if (p->hit_policy != RTMI_HIT_LIST_ORDER) return RTMI_ERR_UNSUPPORTED;
  if (p->search != RTMI_SEARCH_CONE_TREE && p->search != RTMI_SEARCH_LIST_SCAN) return RTMI_ERR_INVALID_ARGUMENT;
  const bool use_tree = p->search == RTMI_SEARCH_CONE_TREE;
  int spp = rtmi_samples_per_pixel(p->pixel_subdivision);
  if (spp < 0) return spp;
  int n1 = 1;
  while (n1 * n1 < spp) ++n1;
  int x0 = p->x0, y0 = p->y0, x1 = p->x1, y1 = p->y1;
  if (x0 == 0 && y0 == 0 && x1 == 0 && y1 == 0) { x1 = p->width - 1; y1 = p->height - 1; }
  if (x0 < 0 || y0 < 0 || x1 > p->width - 1 || y1 > p->height - 1 || x0 > x1 || y0 > y1) return RTMI_ERR_INVALID_ARGUMENT;
  if (p->tile_size < 0 || p->tile_world < 0 || (p->tile_world > 1 && (p->tile_rank < 0 || p->tile_rank >= p->tile_world)))
    return RTMI_ERR_INVALID_ARGUMENT;
  if (p->tile_world > 1 && p->tile_size <= 0) return RTMI_ERR_INVALID_ARGUMENT;

  if (sc->pending) {
    g_last_error = "a frame is already in flight on this scene; call rtmi_render_end first";
    return RTMI_ERR_INVALID_ARGUMENT;
  }
  HIP_TRY(hipSetDevice(sc->device));

  // camera frame: RayTracingFunctions.swift:156-170, WorldCoordinateSequence.swift:58-59
  FrameConst fc{};
  {
    d3 camera = rtx::ld3(cam->pos);
    d3 center = rtx::add(camera, rtx::smul(cam->focal_length, rtx::ld3(cam->dir)));
    d3 u, v;
    rtx::plane_vectors(rtx::ld3(cam->dir), &u, &v);
    d3 hpc = rtx::muls(u, 10.0 / 2.0), vpc = rtx::muls(v, 10.0 / 2.0);
    d3 ul = rtx::add(rtx::sub(center, hpc), vpc);
    d3 ur = rtx::add(rtx::add(center, hpc), vpc);
    d3 ll = rtx::sub(rtx::sub(center, hpc), vpc);
    fc.ul[0] = ul.x; fc.ul[1] = ul.y; fc.ul[2] = ul.z;
    fc.u[0] = u.x; fc.u[1] = u.y; fc.u[2] = u.z;
    fc.v[0] = v.x; fc.v[1] = v.y; fc.v[2] = v.z;
    fc.cam[0] = camera.x; fc.cam[1] = camera.y; fc.cam[2] = camera.z;
    fc.ppw_h = (double)(p->width - 1) / sqrt(rtx::len_squared(rtx::sub(ur, ul)));
    fc.ppw_v = (double)(p->height - 1) / sqrt(rtx::len_squared(rtx::sub(ll, ul)));
  }
  fc.sub = p->pixel_subdivision;
  for (int a = 0; a < 3; ++a) fc.ambient[a] = p->ambient[a];
  fc.W = p->width; fc.H = p->height;
  fc.x0 = x0; fc.y0 = y0; fc.x1 = x1; fc.y1 = y1;
  fc.spp = spp; fc.n1 = n1;
  fc.spp_shift = fc.n1_shift = -1;
  for (int sh = 0; sh < 16; ++sh) {
    if ((1 << sh) == spp) fc.spp_shift = sh;
    if ((1 << sh) == n1) fc.n1_shift = sh;
  }
  fc.tile_size = p->tile_size;
  fc.tiles_x = p->tile_size > 0 ? (p->width - 1 + p->tile_size - 1) / p->tile_size : 1;
  fc.tile_rank = p->tile_rank;
  fc.tile_world = p->tile_world > 1 ? p->tile_world : 1;
  fc.inv_spp = 1.0f / (float)spp;
  {
    size_t need = (size_t)(p->width + p->height) * n1 * 3;
    if (need > sc->cap_tabs) {
      (void)hipFree(sc->d_tabs);
      sc->d_tabs = nullptr;
      sc->cap_tabs = 0;
      HIP_TRY(hipMalloc(&sc->d_tabs, (sizeof(double) + sizeof(float)) * need));  // FP64 tables, then their FP32 copies
      sc->cap_tabs = need;
      sc->tabs_valid = false;
    }
    fc.utab = sc->d_tabs;
    fc.vtab = sc->d_tabs + (size_t)p->width * n1 * 3;
    fc.utab_f = reinterpret_cast<const float*>(sc->d_tabs + sc->cap_tabs);
    fc.vtab_f = fc.utab_f + (size_t)p->width * n1 * 3;
  }

  // 8x8 pixel blocks that contain at least one pixel to render.  The list depends only on the
  // frame geometry and tile selection, so it is rebuilt (and re-uploaded) only when those change.
  const std::array<int, 9> blocks_key = {p->width, p->height, x0, y0, x1, y1, fc.tile_size, fc.tile_rank, fc.tile_world};
  const bool blocks_cached = sc->blocks_valid && sc->blocks_key == blocks_key;
  std::vector<int2>& blocks = sc->h_blocks;
  if (!blocks_cached) {
    blocks.clear();
    sc->blocks_valid = false;
    const bool tiled = fc.tile_world > 1;
    const bool aligned = tiled && (fc.tile_size % BLK == 0);
    for (int by = y0 / BLK; by * BLK < y1; ++by)
      for (int bx = x0 / BLK; bx * BLK < x1; ++bx) {
        if (aligned) {
          int tile = ((by * BLK) / fc.tile_size) * fc.tiles_x + (bx * BLK) / fc.tile_size;
          if (tile % fc.tile_world != fc.tile_rank) continue;
        }
        blocks.push_back(make_int2(bx, by));
      }
  }
  // rendered samples: pixels of the window that this call's tile selection owns
  int64_t live_samples = 0;
  if (fc.tile_world > 1) {
    const int ts = fc.tile_size;
    for (int ty = y0 / ts; ty * ts < y1; ++ty)
      for (int tx = x0 / ts; tx * ts < x1; ++tx) {
        if ((ty * fc.tiles_x + tx) % fc.tile_world != fc.tile_rank) continue;
        int64_t wx = std::min(x1, (tx + 1) * ts) - std::max(x0, tx * ts);
        int64_t wy = std::min(y1, (ty + 1) * ts) - std::max(y0, ty * ts);
        live_samples += wx * wy * spp;
      }
  } else {
    live_samples = (int64_t)(x1 - x0) * (y1 - y0) * spp;
  }
  // With tiles that are whole multiples of the 8x8 blocks the block list already holds exactly the
  // owned pixels, and the kernels need no per-pixel ownership test (two integer divisions per pixel).
  if (fc.tile_world > 1 && fc.tile_size % BLK == 0) {
    fc.tile_world = 1;
    fc.tile_size = 0;
  }
  const size_t samples_per_block = (size_t)BLK_PIX * spp;
  const size_t blocks_per_batch = std::max<size_t>(1, max_batch_samples() / samples_per_block);
  const size_t max_batch_blocks = std::min(blocks_per_batch, std::max<size_t>(blocks.size(), 1));
  int32_t rc = ensure_buffers(sc, max_batch_blocks * samples_per_block, std::max<size_t>(blocks.size(), 1));
  if (rc) return rc;
  sc->dev.n_samples = (int)std::min<size_t>(sc->cap_samples, 0x7fffffff);

  hipStream_t st = sc->stream;
  HIP_TRY(hipEventRecord(sc->ev_a, st));
  HIP_TRY(hipMemsetAsync(sc->d_cnt, 0, sizeof(DevCounters), st));

  // FP32 pre-test of k_gen (scenes without top-level triangle runs: there every sample goes on to
  // the triangle search anyway).  d' = (ul - cam) + uComponent - vComponent in FP32 differs from
  // the FP64 direction by at most rel = sqrt(3)*8*2^-24*B/dmin in relative terms (B bounds the
  // component magnitudes, dmin is the distance from the camera to the image plane), its squared
  // cosine to a unit vector by a few times that; `margin` is far above it.  An object is cleared
  // for a sample only if cos^2 < 1 - (r/|m|)^2 - margin, i.e. the direction misses the sphere
  // around the object by a visible angle; everything closer is decided in FP64 as before.
  fc.n_genf = 0;
  fc.genf = sc->d_genf;
  if (sc->groups[0].n == 0 && sc->dev.n_sphere_like > 0) {
    const double base[3] = {fc.ul[0] - fc.cam[0], fc.ul[1] - fc.cam[1], fc.ul[2] - fc.cam[2]};
    const double dlen = sqrt(cam->dir[0] * cam->dir[0] + cam->dir[1] * cam->dir[1] + cam->dir[2] * cam->dir[2]);
    const double dmin = fabs(cam->focal_length) * dlen;
    const double B = std::max(fabs(base[0]), std::max(fabs(base[1]), fabs(base[2]))) + 20.0;
    const double rel = 1.7320508 * 8.0 * 5.9604644775390625e-08 * B / dmin;
    const double margin = 8.0 * rel + 1.0e-5;
    if (std::isfinite(margin) && margin < 0.05) {
      std::vector<float4> gf;
      for (const rtmi_object& ob : sc->objects) {
        if (ob.kind == RTMI_TRIANGLES) continue;
        const double* c = ob.kind == RTMI_LIGHT ? sc->h_lights[(size_t)ob.first].loc : ob.center;
        const double r = ob.kind == RTMI_LIGHT ? 0.0 : fabs(ob.radius);
        const double m[3] = {c[0] - fc.cam[0], c[1] - fc.cam[1], c[2] - fc.cam[2]};
        const double ml = sqrt(m[0] * m[0] + m[1] * m[1] + m[2] * m[2]);
        float4 e = make_float4(0.f, 0.f, 0.f, -1.f);  // w <= 0: never cleared
        if (std::isfinite(ml) && ml > r * 1.001 && ml > 0) {
          const double c2 = 1.0 - (r / ml) * (r / ml) - margin;
          if (c2 > 1.0e-3) e = make_float4((float)(m[0] / ml), (float)(m[1] / ml), (float)(m[2] / ml), (float)(c2 * (1.0 - 1.0e-6)));
        }
        gf.push_back(e);
      }
      if (gf.size() != sc->genf_uploaded.size() || memcmp(gf.data(), sc->genf_uploaded.data(), sizeof(float4) * gf.size()) != 0) {
        sc->genf_uploaded = gf;  // stays untouched until the next upload: the async copy reads it later
        HIP_TRY(hipMemcpyAsync(sc->d_genf, sc->genf_uploaded.data(), sizeof(float4) * gf.size(), hipMemcpyHostToDevice, st));
      }
      for (int a = 0; a < 3; ++a) fc.base_f[a] = (float)base[a];
      fc.n_genf = (int)gf.size();
    }
  }
  if (!blocks_cached || !sc->blocks_valid) {
    if (!blocks.empty())
      HIP_TRY(hipMemcpyAsync(sc->d_blocks, blocks.data(), sizeof(int2) * blocks.size(), hipMemcpyHostToDevice, st));
    sc->blocks_key = blocks_key;
    sc->blocks_valid = true;
  }

  // Per-frame preparation that depends only on the camera and the frame geometry is kept from the
  // previous frame when those are unchanged (a viewer that re-renders with other lights, or a
  // benchmark loop): the sample tables, and each group's search operands for this camera position.
  {
    std::array<double, 18> key = {fc.ul[0], fc.ul[1], fc.ul[2], fc.u[0], fc.u[1], fc.u[2], fc.v[0], fc.v[1], fc.v[2],
                                  fc.ppw_h, fc.ppw_v, fc.sub, (double)fc.W, (double)fc.H, (double)fc.n1,
                                  (double)(uintptr_t)fc.utab, 0.0, 0.0};
    if (!sc->tabs_valid || memcmp(key.data(), sc->tabs_key.data(), sizeof(double) * key.size()) != 0) {
      int ntab = (p->width + p->height) * n1;
      hipLaunchKernelGGL(k_frame_tables, dim3((ntab + 255) / 256), dim3(256), 0, st, fc, sc->d_tabs,
                         sc->d_tabs + (size_t)p->width * n1 * 3, const_cast<float*>(fc.utab_f), const_cast<float*>(fc.vtab_f));
      sc->tabs_key = key;
      sc->tabs_valid = true;
    }
  }
  d3 camera = rtx::ld3(cam->pos);
  for (auto& g : sc->groups) {
    const int want = use_tree ? 1 : 2;
    if (g.packed_for == want && memcmp(g.packed_cam, cam->pos, sizeof(g.packed_cam)) == 0) continue;
    if ((rc = use_tree ? pack_group_tree(sc, g, camera) : pack_group(sc, g, camera))) return rc;
    g.packed_for = want;
    memcpy(g.packed_cam, cam->pos, sizeof(g.packed_cam));
  }

  // argument blocks of the search launches (one per triangle group), uploaded once per frame
  for (size_t gi = 0; gi < sc->groups.size(); ++gi) {
    const TriGroup& g = sc->groups[gi];
    SearchArgs& a = sc->h_sargs[gi];
    a.queue = gi == 0 ? sc->d_q_top : sc->d_q_child;
    a.nq_ptr = gi == 0 ? &sc->d_cnt->q_top : &sc->d_cnt->q_child;
    a.npacked_ptr = g.d_npacked;
    a.edge = g.d_edge;
    a.packed_tri = g.d_packed_tri;
    a.dirs = sc->d_dirs;
    a.tval = sc->d_tval;
    a.hit_id = sc->d_hit_id;
    a.tri_hit = sc->d_tri_hit;
    a.q_shade = sc->d_q_shade;
    a.cnt = sc->d_cnt;
    a.tris = sc->d_tris;
    a.planes = sc->d_planes;
    a.cam[0] = fc.cam[0]; a.cam[1] = fc.cam[1]; a.cam[2] = fc.cam[2];
    a.mode = gi == 0 ? 1 : 0;
    a.obj_filter = g.object;
    a.group = g.d_group;
    a.best_pos = sc->d_best_pos;
    a.cand = sc->d_cand;
    a.n_group = g.n;
    a.n_samples = (int)std::min<size_t>(sc->cap_samples, 0x7fffffff);
    a.n_tris = (int)sc->n_tris;
    a.n_records = (int)g.n_records;
    a.cand_cap = (int)std::min<size_t>(sc->cap_cand, 0x7fffffff);
    a.tree = g.tree;
  }
  if (sc->sargs_uploaded.size() != sc->h_sargs.size() ||
      memcmp(sc->sargs_uploaded.data(), sc->h_sargs.data(), sizeof(SearchArgs) * sc->h_sargs.size()) != 0) {
    // staged through a copy that stays untouched until the next upload: the async copy reads it later
    sc->sargs_uploaded = sc->h_sargs;
    HIP_TRY(hipMemcpyAsync(sc->d_sargs, sc->sargs_uploaded.data(), sizeof(SearchArgs) * sc->groups.size(), hipMemcpyHostToDevice, st));
  }

  ShadeConst shc;
  for (int a = 0; a < 3; ++a) { shc.cam[a] = fc.cam[a]; shc.ambient[a] = fc.ambient[a]; }

  long long launches = 0;
  int launch_parity = 0;  // tree search launches of the frame alternate between two sets of list counters
  const bool has_top_runs = sc->groups[0].n > 0;
  sc->ev_used = 0;
  auto next_events = [&](hipEvent_t** ev) -> int32_t {
    if (sc->ev_used == sc->ev_pool.size()) {
      std::array<hipEvent_t, 3> t{};
      for (hipEvent_t& e : t) HIP_TRY(hipEventCreate(&e));
      sc->ev_pool.push_back(t);
    }
    *ev = sc->ev_pool[sc->ev_used].data();
    ++sc->ev_used;
    return RTMI_OK;
  };
  bool shade_queue_launched = false;

  for (size_t b0 = 0; b0 < blocks.size(); b0 += blocks_per_batch) {
    size_t nb = std::min(blocks_per_batch, blocks.size() - b0);
    int nsamples = (int)(nb * samples_per_block);
    const int2* d_blk = sc->d_blocks + b0;
    if (b0 > 0) HIP_TRY(hipMemsetAsync(&sc->d_cnt->q_child, 0, sizeof(int) * 3, st));  // the first batch starts from the cleared counters
    int grid = (nsamples + 255) / 256;
    hipLaunchKernelGGL(k_gen, dim3((nsamples + 256 * GEN_ITERS - 1) / (256 * GEN_ITERS)), dim3(256), 0, st, fc, sc->dev, d_blk, nsamples, sc->d_hit_id, sc->d_dirs, sc->d_q_child,
                       sc->d_q_shade, sc->d_q_top, sc->d_cnt);
    const int SR = search_rays_per_lane();
    const int search_grid = (nsamples + 256 * SR - 1) / (256 * SR);
    auto scan_kernel = SR == 8 ? k_search_primary<8> : (SR == 2 ? k_search_primary<2> : k_search_primary<4>);
    auto tree_kernel = SR == 8 ? k_search_tree<8> : (SR == 2 ? k_search_tree<2> : k_search_tree<4>);
    // the triangle search of one group for the rays queued for it
    auto search_group = [&](size_t gi) -> int32_t {
      TriGroup& g = sc->groups[gi];
      hipEvent_t* ev;
      int32_t rc2 = next_events(&ev);
      if (rc2) return rc2;
      HIP_TRY(hipEventRecord(ev[0], st));
      if (use_tree) {
        const int par = launch_parity;
        launch_parity ^= 1;
        hipLaunchKernelGGL(tree_kernel, dim3(search_grid), dim3(256), 0, st, sc->d_sargs + gi, (const f2*)g.d_rec,
                           (const float*)g.d_tedge, (const int*)g.d_slot_pos, par);
        hipLaunchKernelGGL(k_search_exact, dim3(sc->exact_grid), dim3(256), 0, st, sc->d_sargs + gi, par);
        HIP_TRY(hipEventRecord(ev[1], st));
        if (gi == 0)
          hipLaunchKernelGGL(k_search_finish, dim3(grid), dim3(256), 0, st, sc->d_sargs + gi, par);
        else
          hipLaunchKernelGGL(has_top_runs ? k_shade_candidates<true> : k_shade_candidates<false>, dim3(sc->exact_grid), dim3(256),
                             0, st, shc, sc->dev, sc->top_args, (const f2*)sc->d_trec, (const float*)sc->d_gst2,
                             (const int*)sc->groups[0].d_slot_pos, sc->d_sargs + gi, sc->d_rgb, par);
      } else {
        hipLaunchKernelGGL(scan_kernel, dim3(search_grid), dim3(256), 0, st, sc->d_sargs + gi, (const f2*)g.d_cone);
        HIP_TRY(hipEventRecord(ev[1], st));
      }
      HIP_TRY(hipEventRecord(ev[2], st));
      ++launches;
      return RTMI_OK;
    };
    if (sc->groups[0].n > 0) {
      if ((rc = search_group(0))) return rc;
      hipLaunchKernelGGL(k_resolve_top, dim3(grid), dim3(256), 0, st, sc->dev, sc->d_q_top, &sc->d_cnt->q_top, sc->d_tri_hit,
                         sc->d_tri_owner, sc->d_hit_id, sc->d_q_child, sc->d_q_shade, sc->d_cnt);
    }
    for (size_t gi = 1; gi < sc->groups.size(); ++gi)
      if ((rc = search_group(gi))) return rc;
    // the shade queue stays empty when the cone tree shades every triangle hit itself and the scene
    // lists nothing else that can be shaded (no sphere or light object, no top-level triangles)
    if (!(use_tree && !has_top_runs && sc->n_plain_shadeable == 0)) {
      if (b0 == 0) HIP_TRY(hipEventRecord(sc->ev_c, st));  // timed for the first batch only (frames are one batch as a rule)
      hipLaunchKernelGGL(has_top_runs ? k_shade<true> : k_shade<false>, dim3(std::min(grid, 4 * sc->exact_grid)), dim3(256), 0, st, shc, sc->dev, sc->top_args, (const f2*)sc->d_trec,
                         (const float*)sc->d_gst2, (const int*)sc->groups[0].d_slot_pos, sc->d_dirs, sc->d_tval, sc->d_q_shade, &sc->d_cnt->q_shade,
                         sc->d_hit_id, sc->d_rgb, sc->d_cnt);
      if (b0 == 0) HIP_TRY(hipEventRecord(sc->ev_d, st));
      shade_queue_launched = true;
    }
    int npix = (int)(nb * BLK_PIX);
    hipLaunchKernelGGL(k_resolve, dim3((npix + 255) / 256), dim3(256), 0, st, fc, d_blk, npix, sc->d_hit_id, sc->d_rgb,
                       (const rtmi_object*)sc->d_objects, sc->dev.n_objects, sc->d_cnt, (unsigned char*)rgba_dev, (int*)hit_dev);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipEventRecord(sc->ev_b, st));
  HIP_TRY(hipMemcpyAsync(sc->h_cnt, sc->d_cnt, sizeof(DevCounters), hipMemcpyDeviceToHost, st));
  sc->pend_launches = launches;
  sc->pend_live_samples = live_samples;
  sc->pend_shade_timed = shade_queue_launched;
  sc->pend_cam = *cam;
  sc->pend_params = *p;
  sc->pend_rgba = rgba_dev;
  sc->pend_hit = hit_dev;
  sc->pending = true;
  return RTMI_OK;
}

static int32_t render_begin_impl(rtmi_scene* sc, const rtmi_camera* cam, const rtmi_params* p, void* rgba_dev,
                                 void* hit_dev) {
  if (!sc) return RTMI_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> lock(sc->mu);
  return render_enqueue(sc, cam, p, rgba_dev, hit_dev);
}

// Wait for the frame enqueued by render_begin_impl and report its counters.
static int32_t render_end_impl(rtmi_scene* sc, rtmi_counters* counters) {
  if (!sc) return RTMI_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> lock(sc->mu);
  if (!sc->pending) {
    g_last_error = "no frame in flight on this scene";
    return RTMI_ERR_INVALID_ARGUMENT;
  }
  sc->pending = false;
  HIP_TRY(hipSetDevice(sc->device));
  HIP_TRY(hipStreamSynchronize(sc->stream));
  if (sc->h_cnt->overflow && sc->pend_params.search == RTMI_SEARCH_CONE_TREE) {
    // More candidate pairs than the list holds (rays that pass very many triangle cones, e.g. deep
    // stacks of overlapping triangles): the frame is rendered again with the list scan, which
    // needs no such list and gives the same image.
    rtmi_params p2 = sc->pend_params;
    p2.search = RTMI_SEARCH_LIST_SCAN;
    const rtmi_camera cam2 = sc->pend_cam;
    int32_t rc = render_enqueue(sc, &cam2, &p2, sc->pend_rgba, sc->pend_hit);
    if (rc != RTMI_OK) return rc;
    sc->pending = false;
    HIP_TRY(hipStreamSynchronize(sc->stream));
    sc->pend_params.search = RTMI_SEARCH_CONE_TREE;
  }
  const DevCounters& hc = *sc->h_cnt;
  if (hc.bug) {
    // an index read from device memory was out of range (checked() in the kernels replaced it and
    // went on): the frame cannot be trusted
    g_last_error = "internal index check " + std::to_string(hc.bug) + " failed on the device; the frame is not valid";
    return RTMI_ERR_HIP;
  }
  float ms_total = 0.f, ms_search = 0.f, ms_shade = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms_total, sc->ev_a, sc->ev_b));
  for (size_t i = 0; i < sc->ev_used; ++i) {
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, sc->ev_pool[i][0], sc->ev_pool[i][1]));
    ms_search += ms;
    HIP_TRY(hipEventElapsedTime(&ms, sc->ev_pool[i][1], sc->ev_pool[i][2]));
    ms_shade += ms;
  }
  if (sc->pend_shade_timed) {
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, sc->ev_c, sc->ev_d));
    ms_shade += ms;
  }
  if (counters) {
    const int64_t live_samples = sc->pend_live_samples;
    unsigned long long sum[C_NCOUNTERS] = {0};
    for (int sl = 0; sl < NSLOT; ++sl)
      for (int c = 0; c < C_NCOUNTERS; ++c) sum[c] += hc.slot[sl][c];
    const int64_t secondary = (int64_t)(sum[C_REFL] + sum[C_SHADOW]);
    memset(counters, 0, sizeof(*counters));
    counters->samples = live_samples;
    counters->rays_reflection = (int64_t)sum[C_REFL];
    counters->rays_shadow = (int64_t)sum[C_SHADOW];
    // pairs actually evaluated: a ray stops counting at its hit
    counters->tri_tests = (int64_t)(sum[C_TESTS_PRIMARY] + sum[C_TESTS_SECONDARY]);
    counters->sphere_tests = (live_samples + secondary) * sc->dev.n_sphere_like;
    counters->tri_exact_tests = (int64_t)sum[C_EXACT];
    counters->primary_hits = (int64_t)sum[C_HITS];
    counters->shadow_rays_elided = (int64_t)sum[C_SHADOW_ELIDED];
    counters->node_tests = (int64_t)sum[C_NODE_TESTS];
    counters->tri_candidates = (int64_t)hc.cand_total;
    counters->ms_total = ms_total;
    counters->ms_tri_search = ms_search;
    counters->tri_search_launches = sc->pend_launches;
    counters->tri_tests_primary_kernel = (int64_t)sum[C_TESTS_PRIMARY];
    counters->ms_primary_kernel = ms_search;
    counters->ms_shade = ms_shade;
  }
  return RTMI_OK;
}

static int32_t render_impl(rtmi_scene* sc, const rtmi_camera* cam, const rtmi_params* p, void* rgba_dev,
                           void* hit_dev, rtmi_counters* counters) {
  int32_t rc = render_begin_impl(sc, cam, p, rgba_dev, hit_dev);
  if (rc != RTMI_OK) return rc;
  return render_end_impl(sc, counters);
}

extern "C" int32_t rtmi_render_begin(rtmi_scene* sc, const rtmi_camera* cam, const rtmi_params* p, void* rgba_dev,
                                     void* hit_dev) {
  return render_begin_impl(sc, cam, p, rgba_dev, hit_dev);
}

extern "C" int32_t rtmi_render_end(rtmi_scene* sc, rtmi_counters* counters) { return render_end_impl(sc, counters); }

extern "C" int32_t rtmi_render_device(rtmi_scene* sc, const rtmi_camera* cam, const rtmi_params* p, void* rgba_dev,
                                      void* hit_dev, rtmi_counters* counters) {
  return render_impl(sc, cam, p, rgba_dev, hit_dev, counters);
}

extern "C" int32_t rtmi_render(rtmi_scene* sc, const rtmi_camera* cam, const rtmi_params* p, uint8_t* rgba_out,
                               int32_t* hit_id_out, rtmi_counters* counters) {
  if (!sc || !p || !rgba_out) return RTMI_ERR_INVALID_ARGUMENT;
  if (p->width < 2 || p->height < 2) return RTMI_ERR_INVALID_ARGUMENT;
  if (p->width > MAX_FRAME_EDGE || p->height > MAX_FRAME_EDGE) return RTMI_ERR_UNSUPPORTED;
  int spp = rtmi_samples_per_pixel(p->pixel_subdivision);
  if (spp < 0) return spp;
  HIP_TRY(hipSetDevice(sc->device));
  size_t nbytes = (size_t)4 * p->width * p->height;
  size_t nhit = (size_t)p->width * p->height * spp;
  unsigned char* d_rgba = nullptr;
  int* d_hit = nullptr;
  HIP_TRY(hipMalloc(&d_rgba, nbytes));
  hipError_t e = hipMemcpy(d_rgba, rgba_out, nbytes, hipMemcpyHostToDevice);  // unrendered pixels keep the caller's bytes
  if (e == hipSuccess && hit_id_out) {
    e = hipMalloc(&d_hit, sizeof(int) * nhit);
    if (e == hipSuccess) e = hipMemcpy(d_hit, hit_id_out, sizeof(int) * nhit, hipMemcpyHostToDevice);
  }
  int32_t rc = RTMI_OK;
  if (e != hipSuccess) {
    g_last_error = std::string("rtmi_render staging: ") + hipGetErrorString(e);
    rc = e == hipErrorOutOfMemory ? RTMI_ERR_OUT_OF_MEMORY : RTMI_ERR_HIP;
  }
  if (rc == RTMI_OK) rc = render_impl(sc, cam, p, d_rgba, d_hit, counters);
  if (rc == RTMI_OK) {
    e = hipMemcpy(rgba_out, d_rgba, nbytes, hipMemcpyDeviceToHost);
    if (e == hipSuccess && hit_id_out) e = hipMemcpy(hit_id_out, d_hit, sizeof(int) * nhit, hipMemcpyDeviceToHost);
    if (e != hipSuccess) {
      g_last_error = std::string("rtmi_render readback: ") + hipGetErrorString(e);
      rc = RTMI_ERR_HIP;
    }
  }
  (void)hipFree(d_rgba);
  (void)hipFree(d_hit);
  return rc;
}
