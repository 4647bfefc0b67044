// kernels_common.h -- part of rtmi.hip, which includes its parts in a fixed order (each relies on the ones before it).
// Device-side constants and helpers: frame and scene structs, counters, sample <-> pixel mapping.
// Not a stand-alone header.
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

// A pointer that was loaded from a struct in device memory is a "generic" pointer to the compiler, and loads
// through it are flat loads, which count against the LDS/scalar counter as well as the vector-memory one: every
// wait for a scalar load then also waits for all of them.  Every such pointer of this library points to global
// memory; as_global says so.  (The CPU build of the kernels has one address space.)
#if defined(__HIP_DEVICE_COMPILE__)
template <class T>
__device__ __forceinline__ T __attribute__((address_space(1)))* as_global(T* p) {
  return (T __attribute__((address_space(1)))*)p;
}
#else
template <class T>
__device__ inline T* as_global(T* p) { return p; }
#endif

// The lane mask of a predicate.  Votes on a conjunction are written as the AND of the masks of its compares: the
// compiler turns the ballot of a compare into the compare itself (its result is a lane mask in scalar registers),
// whereas the ballot -- or HIP's __any() -- of a combined boolean first makes a per-lane 0/1 of it and compares that
// again, two vector instructions per vote; the walks vote sixteen times per tree record.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ unsigned long long wave_mask(bool p) { return __builtin_amdgcn_ballot_w64(p); }
#else
__device__ inline unsigned long long wave_mask(bool p) { return __ballot(p); }
#endif

// A load of a word that other work-items of the launch may be changing with atomic operations at the same time
// (the value read is one of the values stored; which one must not matter to the caller).
#if defined(__HIP_DEVICE_COMPILE__)
template <class P>
__device__ __forceinline__ int load_relaxed(P p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#else
__device__ inline int load_relaxed(const int* p) { return __atomic_load_n(p, __ATOMIC_RELAXED); }
#endif

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
  // per entry of the block list: 0 = no sample of the block can hit anything (k_block_cull); null = every block is live
  const unsigned char* block_live;
  int nearest;  // hit policy of the frame: 0 = first in list order (the reference), 1 = smallest parameter
};

// Work counters are striped over NSLOT slots (two 64-byte lines per slot) so that the
// per-wave atomic adds of a launch do not serialise on one address; the host sums the slots.
enum { C_REFL = 0, C_SHADOW, C_EXACT, C_HITS, C_TESTS_PRIMARY, C_TESTS_SECONDARY, C_SHADOW_ELIDED, C_NODE_TESTS, C_PRIMARY_TRACED,
       C_FIX_PIXELS,   // pixels whose bytes the FP32 lighting could not settle: evaluated again in FP64 (k_fix_pixels)
       C_EDGE_TESTS,   // cone tree: (ray, triangle) pairs that went through the FP32 edge test
       C_FP32_DECIDED, // single candidates accepted by the FP32 tests alone (inside the triangle and the bounding sphere by a margin)
       C_NCOUNTERS = 16 };
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
  // pixels waiting for k_fix_pixels, in NSLOT lists (workgroup w of k_resolve_fast appends to list w & 63)
  int fix_n[NSLOT][16];
  // the candidates of the rays that have several (SearchArgs::singles), in NSLOT lists like the candidates themselves
  // and with two sets of fill levels like theirs; multi_overflow: a list did not fit (the decisions then scan every candidate)
  int multi_n[2][NSLOT][16];
  int multi_overflow;
  // 8x8 pixel blocks of the batch that k_block_cull_compact found live (all of them, also those beyond the list's capacity);
  // live_overflow: the list did not hold them all -- the frame is incomplete and is rendered again with longer arrays
  int n_live;
  int live_overflow;
  int pad2[13];
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
