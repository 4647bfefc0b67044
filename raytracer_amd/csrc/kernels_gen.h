// kernels_gen.h -- part of rtmi.hip, which includes its parts in a fixed order (each relies on the ones before it).
// K_gen (ray generation and the first sphere-like hit) and k_resolve_top.
// Not a stand-alone header.
// ---------------------------------------------------------------------------------------
// k_gen: one thread per sample of the batch
// ---------------------------------------------------------------------------------------
// Outcome per sample (hit_id): -1 miss / unrendered; >= 0 triangle; <= -2 sphere-like object.
// With no top-level triangle runs the sample is resolved here: a sphere or light hit goes to
// the shade queue, a bounding-sphere hit to the child-search queue.  With top-level runs
// every rendered sample goes on to the top-level triangle search, which resolves it.
#ifndef RTMI_GEN_ITERS
#define RTMI_GEN_ITERS 8
#endif
constexpr int GEN_ITERS = RTMI_GEN_ITERS;  // samples per thread in k_gen; a workgroup covers 2048 consecutive samples

// One thread per 8x8 pixel block: can any sample of the block come near any sphere-like object?  The sample
// positions of a block lie within 4 pixels of its centre in both directions, so their focal-plane points lie
// within rho = 4/pixelsPerWorldHorizontal + 4/pixelsPerWorldVertical of the centre's (u and v are unit vectors,
// WorldCoordinateSequence.swift:62-66), and a direction d = p - camera deviates from the centre's direction dc by
// an angle phi with sin(phi) <= rho/|dc|.  With theta the angle between dc and the unit vector g towards an
// object's centre, no direction of the block makes a smaller angle with g than theta - phi; the block is cleared
// for the object if cos(theta - phi) stays below the cosine the per-sample pre-test of k_gen demands (whose
// margin covers the rounding here, FP64, many times over).  A cleared block is never touched by k_gen and
// written as black by k_resolve.
// The vector from the camera to the centre of block bc on the focal plane, its length, and rho.  False for a patch that
// is wide as seen from the camera (or a NaN): no bound on phi then.
__device__ inline bool block_centre(const FrameConst& fc, int2 bc, double* dx_out, double* dy_out, double* dz_out, double* len_out,
                                    double* rho_out) {
  const double ic = bc.x * BLK + 0.5 * BLK, jc = bc.y * BLK + 0.5 * BLK;
  const double hu = ic / fc.ppw_h, hv = jc / fc.ppw_v;
  const double dx = (fc.ul[0] - fc.cam[0]) + fc.u[0] * hu - fc.v[0] * hv;
  const double dy = (fc.ul[1] - fc.cam[1]) + fc.u[1] * hu - fc.v[1] * hv;
  const double dz = (fc.ul[2] - fc.cam[2]) + fc.u[2] * hu - fc.v[2] * hv;
  const double len = sqrt(dx * dx + dy * dy + dz * dz);
  const double rho = 1.001 * (0.5 * BLK / fabs(fc.ppw_h) + 0.5 * BLK / fabs(fc.ppw_v));
  *dx_out = dx; *dy_out = dy; *dz_out = dz; *len_out = len; *rho_out = rho;
  return len > 4.0 * rho;
}

__device__ inline bool block_is_live(const FrameConst& fc, int2 bc) {
  double dx, dy, dz, len, rho;
  bool live = !block_centre(fc, bc, &dx, &dy, &dz, &len, &rho);  // a patch without a bound is not cleared
  if (!live) {
    const double sphi = rho / len, cphi = sqrt(1.0 - sphi * sphi);
    for (int o = 0; o < fc.n_genf && !live; ++o) {
      const float4 gf = fc.genf[o];
      if (!(gf.w > 0.0f)) { live = true; break; }  // this object is never cleared
      const double ct = ((double)gf.x * dx + (double)gf.y * dy + (double)gf.z * dz) / len;
      const double st = sqrt(fmax(0.0, 1.0 - ct * ct));
      const double cu = ct >= cphi ? 1.0 : ct * cphi + st * sphi;  // largest cosine to g over the block
      live = !(cu < sqrt((double)gf.w));
    }
  }
  return live;
}

__global__ void __launch_bounds__(256)
k_block_cull(FrameConst fc, const int2* __restrict__ blocks, int nblocks, unsigned char* block_live) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nblocks) return;
  block_live[b] = block_is_live(fc, blocks[b]) ? 1 : 0;
}

// The same test for the frames that run without a ray-generation launch, which keep per-sample arrays for the live
// blocks only: every live block gets a slot (in no particular order: one atomic per workgroup of 1024) in live_blocks,
// block_slot[b] is that slot or -1.  Blocks beyond the list's capacity get no slot and raise live_overflow.
constexpr int CULL_BLOCK = 1024;
__global__ void __launch_bounds__(CULL_BLOCK)
k_block_cull_compact(FrameConst fc, const int2* __restrict__ blocks, int nblocks, int2* live_blocks, int* block_slot, int cap_live,
                     DevCounters* cnt) {
  __shared__ int wtot[CULL_BLOCK / 64];
  __shared__ int gbase;
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int2 bc = make_int2(0, 0);
  bool live = false;
  if (b < nblocks) {
    bc = blocks[b];
    live = block_is_live(fc, bc);
  }
  const unsigned long long mask = __ballot(live);
  if (lane == 0) wtot[w] = __popcll(mask);
  __syncthreads();
  if (threadIdx.x == 0) {
    int tot = 0;
    for (int q = 0; q < CULL_BLOCK / 64; ++q) tot += wtot[q];
    gbase = tot ? atomicAdd(&cnt->n_live, tot) : 0;
  }
  __syncthreads();
  if (b >= nblocks) return;
  int slot = -1;
  if (live) {
    slot = gbase + __popcll(mask & ((1ull << lane) - 1ull));
    for (int q = 0; q < w; ++q) slot += wtot[q];
    if (slot < cap_live) {
      live_blocks[slot] = bc;
    } else {
      slot = -1;
      cnt->live_overflow = 1;
    }
  }
  block_slot[b] = slot;
}

__global__ void __launch_bounds__(256)
k_gen(FrameConst fc, SceneDev sc, const int2* __restrict__ blocks, int nsamples, int* hit_id, double* dirs,
      int* q_child, int* q_shade, int* q_top, DevCounters* cnt, double* t_obj) {
  // queue entries are staged in LDS and flushed with one global atomic per queue per workgroup
  __shared__ int lq[3][256 * GEN_ITERS];
  __shared__ int ln[3];
  __shared__ int lbase[3];
  if (threadIdx.x < 3) ln[threadIdx.x] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const bool has_top = sc.n_top_tris > 0;
  unsigned int n_traced = 0;  // samples of this lane whose ray was generated and tested in FP64
  for (int it = 0; it < GEN_ITERS; ++it) {
    const int t = (blockIdx.x * GEN_ITERS + it) * 4 + wv;  // wave-uniform: the 64-sample group of this wave
    const int s = (t << 6) + lane;
    int x = 0, y = 0, kx = 0, ky = 0;
    bool live = false;
    bool dead_block = false;  // wave-uniform
    if ((t << 6) < nsamples) {  // nsamples is a multiple of 64
      int b;
      group_coords(fc, t, &b, &kx, &ky);
      dead_block = fc.block_live && fc.block_live[b] == 0;
      if (!dead_block) live = pixel_coords(fc, blocks, b, lane, &x, &y);
    }
    if (dead_block) continue;  // nothing of this block is written or queued; k_resolve knows
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
      ++n_traced;
      d3 d = sample_dir(fc, x, y, kx, ky);
      int obj;
      if (fc.nearest) {
        double t;
        obj = nearest_sphere_like(sc, rtx::ld3(fc.cam), d, &t);
        // with top-level triangle runs k_resolve_top compares this parameter with the nearest triangle's
        // (t_obj is the colour array, which nothing else uses before shading)
        if (obj >= 0 && has_top) t_obj[(size_t)s * 3] = t;
      } else {
        obj = first_sphere_like(sc, rtx::ld3(fc.cam), d);
      }
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
  wave_count(cnt, C_PRIMARY_TRACED, n_traced);
}

// After the top-level triangle search: combine the first sphere-like object (hit_id from
// k_gen) with the first top-level triangle (tri_hit) by list position.
// A workgroup serves RESOLVE_ITERS x 256 consecutive queue entries and reserves its part of each output queue
// with one atomic: every workgroup of the launch adds to the same two counters, and returning atomics on one
// address are served one after the other.
#ifndef RTMI_RESOLVE_ITERS
#define RTMI_RESOLVE_ITERS 16
#endif
constexpr int RESOLVE_ITERS = RTMI_RESOLVE_ITERS;
__global__ void __launch_bounds__(256)
k_resolve_top(SceneDev sc, const int* __restrict__ q_top, const int* __restrict__ nq_ptr, const int* __restrict__ tri_hit,
              const int* __restrict__ tri_owner, int* hit_id, int* q_child, int* q_shade, DevCounters* cnt,
              const double* __restrict__ tval, const double* __restrict__ t_obj, int nearest) {
  __shared__ int wtot[2][RESOLVE_ITERS * 4];  // per output queue: entries of (round, wave), then their exclusive prefix sums
  __shared__ int gbase[2];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int nq = *nq_ptr;
  int sv[RESOLVE_ITERS];
  unsigned child_bits = 0, shade_bits = 0;
  int rank_child[RESOLVE_ITERS], rank_shade[RESOLVE_ITERS];
#pragma unroll
  for (int it = 0; it < RESOLVE_ITERS; ++it) {
    const int q = (blockIdx.x * RESOLVE_ITERS + it) * 256 + threadIdx.x;
    bool to_child = false, to_shade = false;
    int s = -1;
    if (q < nq) {
      s = checked(q_top[q], sc.n_samples, cnt, 14);
      int id = hit_id[s];
      int th = tri_hit[q];
      if (th >= 0) th = checked(th, sc.n_tris, cnt, 15);
      int obj = id <= -2 ? -2 - id : 0x7fffffff;
      if (th >= 0) {
        const int owner = tri_owner[th];
        bool take = owner < obj;  // list order (also: no sphere-like hit at all)
        if (nearest && id <= -2) {  // the smaller parameter wins, list order among equals
          const double tt = tval[s], to = t_obj[(size_t)s * 3];
          take = tt < to || (tt == to && owner < obj);
        }
        if (take) { id = th; obj = -1; }
      }
      hit_id[s] = id;
      if (id >= 0) to_shade = true;
      else if (id <= -2) {
        if (sc.objects[checked(obj, sc.n_objects, cnt, 16)].kind == RTMI_BOUNDING_SPHERE) to_child = true; else to_shade = true;
      }
    }
    sv[it] = s;
    const unsigned long long mc = __ballot(to_child), ms = __ballot(to_shade);
    const unsigned long long below = (1ull << lane) - 1ull;
    rank_child[it] = __popcll(mc & below);
    rank_shade[it] = __popcll(ms & below);
    if (to_child) child_bits |= 1u << it;
    if (to_shade) shade_bits |= 1u << it;
    if (lane == 0) {
      wtot[0][it * 4 + w] = __popcll(mc);
      wtot[1][it * 4 + w] = __popcll(ms);
    }
  }
  __syncthreads();
  if (threadIdx.x < 128) {  // wave 0 scans the child queue's entries, wave 1 the shade queue's
    const int k = threadIdx.x >> 6;
    static_assert(RESOLVE_ITERS * 4 <= 64, "one wave scans the (round, wave) entries of a queue");
    const int v = lane < RESOLVE_ITERS * 4 ? wtot[k][lane] : 0;
    int incl = v;
    for (int off = 1; off < 64; off <<= 1) {
      const int up = __shfl_up(incl, off);
      if (lane >= off) incl += up;
    }
    if (lane < RESOLVE_ITERS * 4) wtot[k][lane] = incl - v;
    if (lane == 63) gbase[k] = incl ? atomicAdd(k == 0 ? &cnt->q_child : &cnt->q_shade, incl) : 0;
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < RESOLVE_ITERS; ++it) {
    if ((child_bits >> it) & 1u) q_child[gbase[0] + wtot[0][it * 4 + w] + rank_child[it]] = sv[it];
    if ((shade_bits >> it) & 1u) q_shade[gbase[1] + wtot[1][it * 4 + w] + rank_shade[it]] = sv[it];
  }
}
