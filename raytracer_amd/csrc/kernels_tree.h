// kernels_tree.h -- part of rtmi.hip, which includes its parts in a fixed order (each relies on the ones before it).
// The cone tree search (RTMI_SEARCH_CONE_TREE): k_search_tree, k_search_exact, k_search_finish.
// Not a stand-alone header.
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
constexpr int MULTI_STAGE = 128; // the same for the candidates of rays with several (one step adds at most 2 x 64)
// Nearest-hit policy: k_search_exact marks the candidates that are hits with this bit of their list position
// (list positions stay below it: the host refuses larger groups) and lowers the ray's smallest parameter;
// k_search_nearest then lets the marked candidates whose parameter IS that smallest one compete by position.
constexpr int NEAR_MARK = 1 << 30;

__device__ __forceinline__ unsigned long long f64_bits(double v) {
  unsigned long long b;
  __builtin_memcpy(&b, &v, 8);
  return b;
}

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
    cnt->multi_n[parity ^ 1][threadIdx.x][0] = 0;
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

// ---------------------------------------------------------------------------------------
// the search without a ray-generation launch (see search_tree_body<R, true>)
// ---------------------------------------------------------------------------------------
constexpr int SURE_BIT = 1 << 30;  // in a candidate's list position: the FP32 edge test puts the ray inside the triangle by a margin

// Per frame and top-level object (all of them bounding spheres here): Sphere.intersections (Renderables.swift:248-282)
// with a = -d.cte and delta = a^2 - k answers "hit" for certain when a and delta exceed the FP32 evaluation error.
struct SphereF {
  float cx, cy, cz;  // camera - centre (the FP64 difference, rounded once)
  float k;           // |camera - centre|^2 - radius^2 (FP64, rounded)
  float ea;          // bound on |a - a*|: (dir_err + 8 * 2^-24) * |camera - centre|_1
  float kabs;        // |k|
  float pad0, pad1;
};

struct FastArgs {
  const int2* blocks;   // the batch's LIVE 8x8 pixel blocks (k_block_cull_compact): the per-sample arrays are indexed by them
  int nsamples;         // live blocks x 64 x samples per pixel; set by each kernel from the device's count (fast_live_counts)
  int cap_live;         // blocks the live list and the per-sample arrays have room for
  const int2* all_blocks;  // the batch's blocks as the host listed them, and per such block its slot among the live ones or -1
  const int* block_slot;
  int obj;              // list index of the bounding sphere whose children this launch searches
  int init_hit;         // this launch writes hit_id = -1 for every rendered sample of the batch (the first group's launch)
  float dir_err;        // bound on |d - d*| for the FP32 directions of fast_dir (host: fast_dir_err)
  int n_materials;
  int* hit_id;
  float4* rgbf;               // per sample: colour and its error bound (rtx32::Shade32)
  const float4* pos_plane;    // per list position of the group: unit normal, c - n.camera (k_pack_tree)
  const int2* pos_mt;         // per list position: the triangle's material index and its index in the scene's triangle array
  int ns_uniform;             // the specular exponent if every material of the scene has the same one, else 0
  const rtx32::LightF* lights;
  const rtx32::MatF* mats;
  const SphereF* spheres;
  int* fix_list;              // pixels for k_fix_pixels: NSLOT lists of fix_cap / NSLOT entries
  int fix_cap;
  int force;                  // tests: bit 0 = no pair is accepted by the FP32 margins, bit 1 = every pixel goes to k_fix_pixels
  const int* entries;         // per live block ENTRY_K records where the walks of its waves start (k_block_entries); null: at the root
  int entry_masks;            // the entries of leaf records carry a mask of the triangles a ray of the block can reach
};

// Where the walk of a block's waves starts.  Every direction of an 8x8 pixel block lies within a small angle of the
// direction to the block's centre, so the upper records of the tree give the same answer for all waves of the block --
// and mostly "one or two children" -- which one work-item per block finds out beforehand (k_block_entries) instead of
// every lane of every wave: the records a wave would enter from the root are replaced by their children that the block
// can reach, as long as at most ENTRY_K records result.  An entry is (level << ENTRY_SHIFT) | record index, -1 = none.
constexpr int ENTRY_K = 8;
constexpr int ENTRY_SHIFT = 27;
constexpr int ENTRY_FINAL = 1 << 30;  // (inside k_block_entries only) this entry is not expanded any further
// A leaf record's entry may name the triangles of the record that a ray of the block can reach (the same test, applied
// to the record's eight triangle cones): record index in the low ENTRY_LEAF_BITS bits, the mask above them.
constexpr int ENTRY_LEAF_BITS = 19;
static_assert(ENTRY_K == 8, "k_block_entries keeps a block's list in the eight lanes of a group");

// Every kernel of the path starts with this: the number of live blocks is only known on the device.
__device__ __forceinline__ void fast_live_counts(FastArgs& fa, const FrameConst& fc, const DevCounters* cnt) {
  fa.nsamples = min(max(cnt->n_live, 0), fa.cap_live) * BLK_PIX * fc.spp;
}

// The FP32 direction of sub-sample (kx, ky) of pixel (x, y): pixelToWorldCoordinate (WorldCoordinateSequence.swift:62-66)
// minus the camera, from the FP32 copies of the per-column and per-row tables, normalised with v_rsq_f32.  *f2_out gets
// the squared length before normalisation, fx.. the unnormalised vector (the pre-test of the caller wants them).
__device__ __forceinline__ void fast_dir(const FrameConst& fc, int x, int y, int kx, int ky, float* dx, float* dy, float* dz,
                                         float* fx_out, float* fy_out, float* fz_out, float* dd_out) {
  const float* uc = as_global(fc.utab_f) + (size_t)(x * fc.n1 + kx) * 3;
  const float* vc = as_global(fc.vtab_f) + (size_t)(y * fc.n1 + ky) * 3;
  const float fx = fc.base_f[0] + uc[0] - vc[0], fy = fc.base_f[1] + uc[1] - vc[1], fz = fc.base_f[2] + uc[2] - vc[2];
  const float dd = fx * fx + fy * fy + fz * fz;
  const float inv = rtx32::rsq(dd);
  *dx = fx * inv; *dy = fy * inv; *dz = fz * inv;
  *fx_out = fx; *fy_out = fy; *fz_out = fz; *dd_out = dd;
}

// Pixel and sub-sample of sample s of the batch (s = ((block * spp + k) << 6) + pixel of the 8x8 block).
__device__ __forceinline__ void fast_sample_coords(const FrameConst& fc, const int2* __restrict__ blocks, int s, int* x, int* y,
                                                   int* kx, int* ky) {
  int b;
  group_coords(fc, s >> 6, &b, kx, ky);
  const int2 bc = blocks[b];
  const int pix = s & (BLK_PIX - 1);
  *x = bc.x * BLK + (pix & (BLK - 1));
  *y = bc.y * BLK + (pix >> 3);
}

// The R rays of a lane in the search without a ray-generation launch.  A sample whose pixel is not rendered, whose
// block k_block_cull cleared, or whose direction stays clear of the bounding sphere's cone (the pre-test that k_gen
// applies in the other path, same operands) gets sid = -1 and a NaN direction (blocks that k_block_cull_compact cleared
// are not in the list).  Returns false if the wave has no ray.
template <int R>
__device__ __forceinline__ bool fast_prologue(const FrameConst& fc, const FastArgs& fa, DevCounters* cnt, int wave, int lane,
                                              int (&sid)[R], float (&dx)[R], float (&dy)[R], float (&dz)[R]) {
  const float qnan = __builtin_nanf("");
  const float4 gf = fc.genf[fa.obj];
  bool any = false;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int t = wave * R + r;  // wave-uniform: a group of 64 samples = the 64 pixels of one block at one sub-sample
    sid[r] = -1;
    dx[r] = dy[r] = dz[r] = qnan;
    if ((t << 6) >= fa.nsamples) continue;
    int b, kx, ky;
    group_coords(fc, t, &b, &kx, &ky);
    int x, y;
    if (!pixel_coords(fc, fa.blocks, b, lane, &x, &y)) continue;
    const int s = (t << 6) + lane;
    if (fa.init_hit) fa.hit_id[s] = -1;
    float ux, uy, uz, fx, fy, fz, dd;
    fast_dir(fc, x, y, kx, ky, &ux, &uy, &uz, &fx, &fy, &fz, &dd);
    const float qd = gf.x * fx + gf.y * fy + gf.z * fz;
    const bool maybe = gf.w <= 0.0f || (qd > 0.0f && !(qd * qd < gf.w * dd));
    if (maybe) {
      sid[r] = s;
      dx[r] = ux; dy[r] = uy; dz[r] = uz;
      any = true;
    }
  }
  return __any(any);
}

// The same for the wave's staged candidates of rays that have several.  A list that does not fit only costs time: the
// decisions then scan every candidate (k_search_exact_multi).
__device__ __forceinline__ void multi_flush(const SearchArgs* ap, const int2* mstage, int fill, int lane, int list, int parity) {
  if (fill == 0) return;
  DevCounters* cnt = ap->cnt;
  const int list_cap = ap->multi_cap / NSLOT;
  int base = 0;
  if (lane == 0) base = atomicAdd(&cnt->multi_n[parity][list][0], fill);
  base = __builtin_amdgcn_readfirstlane(base);
  if (base + fill > list_cap || base < 0) {
    if (lane == 0) cnt->multi_overflow = 1;
    return;
  }
  int2* dst = ap->multi + (size_t)list * list_cap + base;
  for (int i = lane; i < fill; i += 64) dst[i] = mstage[i];
}

// Each wave walks the tree depth first for its R*64 rays.  The walk is wave-uniform: `lvl`, the
// record index `ri` and `masks` (one byte of children still to visit per level) live in scalar
// registers.  Record (l, r) holds the cones of nodes 8r..8r+7 of level l; the children of node
// (l, j) are record (l-1, j); level-0 nodes are triangle slots, and a level-0 record is not
// pushed but evaluated per lane: every (ray, slot) pair that passes becomes a candidate.
// FAST (scenes whose top-level list holds nothing but bounding spheres without a material, pixels only): there is no
// ray-generation launch before this one.  The wave's R*64 rays are the samples base .. base + R*64 - 1 of the batch; a
// lane derives its rays' directions in FP32 from the per-column and per-row tables (fast_prologue), and the candidates
// carry a mark when the ray lies inside the triangle by the same margin that the filter allows outside it.
template <int R, bool FAST>
__device__ __forceinline__ void search_tree_body(const SearchArgs* __restrict__ ap, const f2* __restrict__ rec,
                                                 const float* __restrict__ edge_all, const int* __restrict__ slot_pos, int parity,
                                                 const FrameConst* fcp, const FastArgs* fap, int wave) {
  __shared__ int2 stage_all[4][CAND_STAGE];
  __shared__ int2 mstage_all[4][MULTI_STAGE];
  const int nq = FAST ? fap->nsamples : *ap->nq_ptr;
  const int lane = threadIdx.x & 63;
  const int base = wave * 64 * R;
  if (base >= nq) return;  // no workgroup-wide synchronisation below: waves leave on their own
  int2* stage = stage_all[threadIdx.x >> 6];
  int2* mstage = mstage_all[threadIdx.x >> 6];
  const bool singles = ap->singles != 0;
  int mfill = 0;

  int sid[R];
  float dx[R], dy[R], dz[R];
  if constexpr (FAST) {
    if (!fast_prologue<R>(*fcp, *fap, ap->cnt, wave, lane, sid, dx, dy, dz)) return;  // wave-uniform: none of its samples can hit anything
  } else {
    search_prologue<R>(ap, base, nq, lane, sid, dx, dy, dz);
  }
  {
    int* best_pos = ap->best_pos;  // this launch decides these rays: no hit so far
    const bool nearest = ap->nearest != 0;
    double* tval = ap->tval;  // nearest-hit policy: the smallest parameter so far, as an unsigned 64-bit minimum
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (sid[r] >= 0) {
        best_pos[sid[r]] = 0x7fffffff;
        if (nearest) tval[sid[r]] = __builtin_inf();
      }
  }
  unsigned n_leaf_tri = 0, n_node_rec = 0, n_edge_tri = 0;
  int fill = 0;
  // per ray of the lane: -1 no candidate so far, -2 several, else the list position of its only one (SearchArgs::singles)
  int pos1[R];
#pragma unroll
  for (int r = 0; r < R; ++r) pos1[r] = -1;

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
  auto leaf = [&](int ri, unsigned tmask) {  // tmask (wave-uniform): the triangles of the record to look at
    if ((unsigned)(ri * 8) >= (unsigned)n_slots) {  // only a defective record can lead here
      ap->cnt->bug = 3;
      return;
    }
    n_leaf_tri += (FAST && R == 4) ? __popc(tmask) : 8;
    const f2* rp = rec + (size_t)(o0 + ri) * (REC_DWORDS / 2);
    f2 cur[12];
#pragma unroll
    for (int c = 0; c < 12; ++c) cur[c] = rp[c];
    const unsigned always = reinterpret_cast<const unsigned*>(rp)[REC_ALWAYS];  // wide triangles (NaN cones)
    const float* edge = edge_all + (size_t)ri * (8 * EDGE_FLOATS);
#pragma unroll
    for (int pr = 0; pr < 4; ++pr) {
      if constexpr (FAST && R == 4) {  // (the other instantiations are never given a mask: see the host)
        if (!((tmask >> (2 * pr)) & 3u)) continue;
      }
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
        if constexpr (FAST && R == 4) {
          if (!((tmask >> jj) & 1u)) continue;
        }
        float mq = h ? q[0].y : q[0].x;  // fmaxf drops the NaN of absent rays
#pragma unroll
        for (int r = 1; r < R; ++r) mq = __builtin_fmaxf(mq, h ? q[r].y : q[r].x);
        if (!__any(mq >= 1.0f) && !((always >> jj) & 1u)) continue;
        ++n_edge_tri;
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
          if (pass) {
            // FAST: e <= -1 puts the ray inside all three edge planes by the margin that covers the FP32 evaluation
            int mark = 0;
            if constexpr (FAST) mark = e <= -1.0f ? SURE_BIT : 0;
            stage[fill + __popcll(bm & ((1ull << lane) - 1ull))] = make_int2(sid[r], pos | mark);
          }
          fill += n;
          // a ray's second candidate takes its first one along to the list of candidates that need a decision among
          // themselves; later ones join it
          const int prev = pos1[r];
          const bool again = pass && prev != -1;
          if (pass) pos1[r] = prev == -1 ? pos : -2;
          const unsigned long long m2 = singles ? __ballot(again) : 0ull;
          if (m2) {
            const bool first_too = again && prev >= 0;
            const unsigned long long m1 = __ballot(first_too);
            const int n2 = __popcll(m2), add = n2 + __popcll(m1);
            if (mfill + add > MULTI_STAGE) {
              multi_flush(ap, mstage, mfill, lane, wave & (NSLOT - 1), parity);
              mfill = 0;
              __builtin_amdgcn_wave_barrier();
            }
            const unsigned long long below = (1ull << lane) - 1ull;
            if (again) mstage[mfill + __popcll(m2 & below)] = make_int2(sid[r], pos);
            if (first_too) mstage[mfill + n2 + __popcll(m1 & below)] = make_int2(sid[r], prev);
            mfill += add;
          }
        }
      }
    }
  };

  // where the walk starts: at the root record(s), or at the records k_block_entries found for the block of this wave's
  // samples (FAST, and the wave's R groups of 64 samples belong to one block)
  const int* entries = nullptr;
  if constexpr (FAST) {
    if (fap->entries) {
      // (the wave's number is the same in all its lanes, which the compiler cannot know: without the readfirstlane the
      // whole walk below would count as divergent control flow and keep its state in vector registers)
      const int wave_u = __builtin_amdgcn_readfirstlane(wave);
      int b0, b1, kx, ky;
      const int t_last = min(wave_u * R + R - 1, (nq >> 6) - 1);
      group_coords(*fcp, wave_u * R, &b0, &kx, &ky);
      group_coords(*fcp, t_last, &b1, &kx, &ky);
      b0 = __builtin_amdgcn_readfirstlane(b0);
      b1 = __builtin_amdgcn_readfirstlane(b1);
      if (b0 == b1 && (unsigned)b0 < (unsigned)fap->cap_live) entries = fap->entries + (size_t)b0 * ENTRY_K;
    }
  }
  const int n_start = entries ? ENTRY_K : n_root;
#pragma nounroll
  for (int t = 0; t < n_start; ++t) {
    // a start is a record (top, ri) with top >= 1, or a leaf record: the latter enters the loop below as the only
    // pending child of its (unevaluated) parent, so that the leaf code exists once
    int top = ltop, ri = t;
    unsigned start_mask = 0xffu;  // of a start that is a leaf record: its triangles to look at
    if (entries) {
      const int e = __builtin_amdgcn_readfirstlane(entries[t]);
      if (e < 0) break;
      top = (e >> ENTRY_SHIFT) & 7;
      ri = e & ((1 << ENTRY_SHIFT) - 1);
      if constexpr (FAST && R == 4) {
        if (top == 0 && fap->entry_masks) {
          start_mask = (unsigned)(ri >> ENTRY_LEAF_BITS) & 0xffu;
          if (start_mask == 0) start_mask = 0xffu;  // (an entry k_block_entries did not get round to)
          ri &= (1 << ENTRY_LEAF_BITS) - 1;
        }
      }
      if (top > ltop) {  // only a defective entry can lead here
        ap->cnt->bug = 6;
        break;
      }
    }
    int lvl;
    unsigned long long masks;
    if (top == 0) {
      masks = 1ull << (8 + (ri & 7));
      ri >>= 3;
      top = lvl = 1;
    } else {
      lvl = top;
      unsigned m = tree_eval<R>(rec + (size_t)checked(level_off(lvl) + ri, n_records, ap->cnt, 5) * (REC_DWORDS / 2), dx, dy, dz);
      masks = (unsigned long long)m << (8 * lvl);
      ++n_node_rec;
    }
    for (;;) {
      const unsigned cur = (unsigned)(masks >> (8 * lvl)) & 0xffu;
      if (cur == 0) {
        if (lvl == top) break;
        ++lvl;
        ri >>= 3;
        continue;
      }
      const int c = __builtin_ctz(cur);
      masks &= ~(1ull << (8 * lvl + c));
      const int child = ri * 8 + c;
      if (lvl == 1) {
        leaf(child, start_mask);
      } else {
        --lvl;
        ri = child;
        unsigned m = tree_eval<R>(rec + (size_t)checked(level_off(lvl) + ri, n_records, ap->cnt, 5) * (REC_DWORDS / 2), dx, dy, dz);
        masks |= (unsigned long long)m << (8 * lvl);
        ++n_node_rec;
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
  cand_flush(ap, stage, fill, lane, wave & (NSLOT - 1), parity);
  multi_flush(ap, mstage, mfill, lane, wave & (NSLOT - 1), parity);
  if (singles) {
    int* best_pos = ap->best_pos;
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (sid[r] >= 0 && pos1[r] >= 0) best_pos[sid[r]] = ~pos1[r];
  }

  // work counters: (ray, triangle) cone tests at leaf level, (ray, node) tests above it
  int active = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) active += __popcll(__ballot(sid[r] >= 0));
  DevCounters* cnt = ap->cnt;
  wave_count(cnt, C_TESTS_PRIMARY, lane == 0 ? (unsigned long long)n_leaf_tri * (unsigned long long)active : 0ull);
  wave_count(cnt, C_NODE_TESTS, lane == 0 ? 8ull * n_node_rec * (unsigned long long)active : 0ull);
  wave_count(cnt, C_EDGE_TESTS, lane == 0 ? (unsigned long long)n_edge_tri * (unsigned long long)active : 0ull);
  if constexpr (FAST) wave_count(cnt, C_PRIMARY_TRACED, lane == 0 ? (unsigned long long)active : 0ull);
}

#ifndef RTMI_TREE_WAVES
#define RTMI_TREE_WAVES 7
#endif
#ifndef RTMI_TREE_WAVES_GENERAL
#define RTMI_TREE_WAVES_GENERAL 6
#endif
template <int R>
__global__ void __launch_bounds__(256, R <= 4 ? RTMI_TREE_WAVES_GENERAL : 6)
k_search_tree(const SearchArgs* __restrict__ ap, const f2* __restrict__ rec, const float* __restrict__ edge_all,
              const int* __restrict__ slot_pos, int parity) {
  search_tree_body<R, false>(ap, rec, edge_all, slot_pos, parity, nullptr, nullptr, (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
}

template <int R>
__global__ void __launch_bounds__(256, R <= 4 ? RTMI_TREE_WAVES : 6)
k_search_tree_fast(const SearchArgs* __restrict__ ap, const f2* __restrict__ rec, const float* __restrict__ edge_all,
                   const int* __restrict__ slot_pos, int parity, FrameConst fc, FastArgs fa) {
  // The grid covers the samples the per-sample arrays have room for; how many of them exist only the device knows, and the
  // waves beyond them leave at once.  (A grid of resident workgroups that loop over the groups of samples was measured
  // 5 to 20 % slower: the hardware's own dispatch balances waves of unequal work better than a fixed stride does.)
  fast_live_counts(fa, fc, ap->cnt);
  search_tree_body<R, true>(ap, rec, edge_all, slot_pos, parity, &fc, &fa, (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
}

// One work-item per live block: the records at which the walks of the block's waves start (see ENTRY_K).
// A child of a record can be reached by a ray of the block only if the direction c to the block's centre makes an angle
// of at most theta + Phi with the child's axis a, where theta is the half-angle the child's stored vector stands for
// (w = a / cos(theta): the walk passes a ray iff d.w >= 1) and Phi bounds the angle between c and any direction the walk
// may test for the block: phi of block_is_live, plus the distance between the FP32 directions and the reference's.
// With |w| = 1 / cos(theta):  cos(angle(c, a)) >= cos(theta + Phi)  <=>  c.w - cos(Phi) >= -sin(Phi) sqrt(|w|^2 - 1),
// which is decided without a root or a division: q = c.w - cos(Phi) is non-negative, or q^2 <= sin^2(Phi) (|w|^2 - 1).
// Everything here is FP64 on the walk's own FP32 operands, with slack of 1e-6 relative and 1e-9 absolute; the test
// is only ever more permissive than the walk's.  A block without a bound on phi starts at the root.
// Eight lanes per block: lane j of the group holds entry j of the block's list and tests child j of the record that is
// being expanded.  All groups of a wave step together (the cross-lane operations are wave-wide); a group that is done
// idles until the wave's last one is.
// blocks, n_blocks: the blocks to look at; n_blocks < 0: the live list of k_block_cull_compact, whose length (at most cap)
// only the device knows.  entries_out (may be null): ENTRY_K entries per block.  live_out (may be null): per block 1 if
// any record is left that a ray of the block can reach, else 0 -- no ray of the block can hit a triangle of the group.
__global__ void __launch_bounds__(256)
k_block_entries(FrameConst fc, const int2* __restrict__ blocks, int n_blocks, int cap, float dir_err, const f2* __restrict__ rec_all, TreeDev T,
                int n_records, DevCounters* cnt, int* entries_out, unsigned char* live_out, int leaf_masks) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int slot = tid >> 3, j = tid & 7;
  const int lane = threadIdx.x & 63, g0 = lane & ~7;  // first lane of the group
  const int n_blk = n_blocks >= 0 ? n_blocks : min(max(cnt->n_live, 0), cap);
  const bool exists = slot < n_blk;
  int e = -1;  // this lane's entry
  double cx = 0.0, cy = 0.0, cz = 0.0, sPhi = 1.0, cPhi = 0.0;
  if (exists) {
    double dx, dy, dz, len, rho;
    const bool bounded = block_centre(fc, blocks[slot], &dx, &dy, &dz, &len, &rho);
    sPhi = rho / len + 2.0 * (double)dir_err + 2.0e-6;
    const bool usable = T.n_root == 1 && bounded && sPhi < 0.5 && T.ltop >= 1;
    cPhi = sqrt(fmax(0.0, 1.0 - sPhi * sPhi));
    cx = dx / len; cy = dy / len; cz = dz / len;
    if (j == 0) e = (T.ltop << ENTRY_SHIFT) | (usable ? 0 : ENTRY_FINAL);
  }
  const float* rec = reinterpret_cast<const float*>(rec_all);
  for (int iter = 0; iter < 8 * ENTRY_K; ++iter) {
    // the group's entry of the highest level that may still be expanded: the largest key wins
    // (with leaf_masks a leaf record's entry is open too, until its triangles have been looked at: last of all)
    const int lvl_e = (e >> ENTRY_SHIFT) & 7;
    const bool open = e >= 0 && !(e & ENTRY_FINAL) && (lvl_e >= 1 || leaf_masks);
    int key = open ? ((lvl_e << 3) | j) : -1;
    for (int sh = 1; sh < 8; sh <<= 1) key = max(key, __shfl_xor(key, sh));
    if (!__any(key >= 0)) break;  // every group of the wave is done
    const bool active = key >= 0;
    const int pick = active ? (key & 7) : 0;
    const int pe = __shfl(e, g0 + pick);
    const int pick_lvl = (pe >> ENTRY_SHIFT) & 7;
    const int ri = pe & ((1 << ENTRY_SHIFT) - 1);
    bool pass = false;
    if (active) {
      const int rix = checked(T.rec_off[pick_lvl] + ri, n_records, cnt, 7);
      const float* rp = rec + (size_t)rix * REC_DWORDS;
      const unsigned always = reinterpret_cast<const unsigned*>(rp)[REC_ALWAYS];
      const float* w = rp + (j >> 1) * 6 + (j & 1);
      const double wx = w[0], wy = w[2], wz = w[4];
      const double wl2 = wx * wx + wy * wy + wz * wz;
      if (wl2 != wl2 || ((always >> j) & 1u)) {
        pass = true;  // a wide child: always entered
      } else if (wl2 > 0.0) {  // (an absent or culled child has the zero vector: never entered)
        const double slack = 1.0e-9 * (1.0 + wl2);
        const double q = (cx * wx + cy * wy + cz * wz) - cPhi;
        pass = q >= -slack || q * q <= sPhi * sPhi * fmax(wl2 - 1.0, 0.0) * 1.000001 + slack;
      }
    }
    const unsigned long long pb = __ballot(pass), eb = __ballot(e >= 0);
    const unsigned mask = (unsigned)(pb >> g0) & 0xffu;  // children to enter
    const unsigned have = (unsigned)(eb >> g0) & 0xffu;  // entries held
    const int k = __popc(mask), n = __popc(have);
    // a leaf record (leaf_masks): its entry makes way for one that carries the mask of the triangles that can be reached
    // -- or for none, if there is no such triangle
    const bool is_leaf = active && pick_lvl == 0;
    const int k_new = is_leaf ? (mask ? 1 : 0) : k;
    const bool expand = active && (is_leaf || n - 1 + k <= ENTRY_K);
    // the new list: the entries kept (all but the picked one), then the children
    const unsigned kept = expand ? (have & ~(1u << pick)) : have;
    const int n_kept = __popc(kept);
    // source lane of the entry that moves to place j: the j-th set bit of `kept`
    int src = 0, seen = 0, child_bit = 0, seen_c = 0;
    for (int q = 0; q < 8; ++q) {
      if ((kept >> q) & 1u) { if (seen == j) src = q; ++seen; }
      if ((mask >> q) & 1u) { if (seen_c == j - n_kept) child_bit = q; ++seen_c; }
    }
    const int moved = __shfl(e, g0 + src);
    if (active && !expand) {
      if (j == pick) e |= ENTRY_FINAL;
    } else if (expand) {
      if (j < n_kept) {
        e = moved;
      } else if (j < n_kept + k_new) {
        const long long child = (long long)ri * 8 + child_bit;
        if (is_leaf) e = ((int)mask << ENTRY_LEAF_BITS) | ri | ENTRY_FINAL;
        else if (child >= (1ll << ENTRY_SHIFT)) { cnt->bug = 8; e = -1; }  // cannot happen: the host leaves such trees to the root
        else e = ((pick_lvl - 1) << ENTRY_SHIFT) | (int)child;
      } else {
        e = -1;
      }
    }
  }
  const unsigned left = (unsigned)(__ballot(e >= 0) >> g0) & 0xffu;
  if (exists) {
    if (entries_out) entries_out[(size_t)slot * ENTRY_K + j] = e >= 0 ? (e & ~ENTRY_FINAL) : -1;
    if (live_out && j == 0) live_out[slot] = left ? 1 : 0;
  }
}

// One lane per candidate: the reference's own arithmetic in FP64; a hit competes for the ray's
// lowest list position.  Workgroup w serves candidate list w & 63 together with the other
// workgroups of that residue (the grid is a multiple of 64 workgroups).
template <bool NEAREST>
__global__ void __launch_bounds__(256)
k_search_exact(const SearchArgs* __restrict__ ap, int parity) {
  const DevCounters* cnt = ap->cnt;
  unsigned int n_exact = 0;
  if (!cnt->overflow) {  // after an overflow the lists are incomplete and the frame is rendered again
    const int list = blockIdx.x & (NSLOT - 1), part = blockIdx.x / NSLOT, parts = gridDim.x / NSLOT;
    const int list_cap = ap->cand_cap / NSLOT;
    const int n = min(cnt->cand_n[parity][list][0], list_cap);
    auto* cand = as_global(ap->cand) + (size_t)list * list_cap;
    const auto* dirs = as_global(ap->dirs);
    const auto* tris = as_global(ap->tris);
    const auto* planes = as_global(ap->planes);
    auto* best_pos = as_global(ap->best_pos);
    auto* tbits = as_global(reinterpret_cast<unsigned long long*>(ap->tval));
    constexpr bool nearest = NEAREST;
    const d3 o = rtx::ld3(ap->cam);
    // A step begins with a chain of dependent loads: the entry, the triangle index it leads to, the triangle.
    // They are spread over three steps -- this step's triangle, the next step's triangle index and the entry of the
    // step after that are requested together -- so that a step waits for one memory latency instead of three.
    const int stride = parts * blockDim.x;
    const auto* group = as_global(ap->group);
    int i = part * blockDim.x + threadIdx.x;
    DevCounters* wcnt = ap->cnt;
    const int n_group = ap->n_group, n_samples = ap->n_samples, n_tris = ap->n_tris;
    auto entry = [&](int k) {  // candidate k with both indices range-checked
      int2 e = cand[k];
      e.x = checked(e.x, n_samples, wcnt, 6);
      e.y = checked(e.y, n_group, wcnt, 7);
      return e;
    };
    int2 c = i < n ? entry(i) : make_int2(0, 0);             // this step's entry
    int2 c1 = i + stride < n ? entry(i + stride) : c;        // the next step's
    int ti = i < n ? checked(group[c.y], n_tris, wcnt, 8) : 0;
    while (i < n) {
      const int i2 = i + 2 * stride;
      const int s = c.x, pos = c.y;
      // requests of this step, none depending on another
      const auto* dp = dirs + (size_t)s * 3;
      const auto* tv0 = &tris[ti].v[0][0];  // nine doubles: the three vertices
      const auto* plp = planes + ti;
      const double d0 = dp[0], d1 = dp[1], d2 = dp[2];
      const double v0 = tv0[0], v1 = tv0[1], v2 = tv0[2], v3 = tv0[3], v4 = tv0[4], v5 = tv0[5], v6 = tv0[6], v7 = tv0[7], v8 = tv0[8];
      TriPlane pl;
      pl.nx = plp->nx; pl.ny = plp->ny; pl.nz = plp->nz; pl.c = plp->c;
      const int ti1_raw = group[c1.y];
      int2 c2 = c1;
      if (i2 < n) c2 = cand[i2];
      d3 P;
      double tv;
      n_exact += 1;
      if (rtx::tri_hit(rtx::mk(v0, v1, v2), rtx::mk(v3, v4, v5), rtx::mk(v6, v7, v8), pl, o, rtx::mk(d0, d1, d2), &tv, &P)) {
        if (nearest) {  // t > 0: the bit patterns of positive doubles order like the values
          atomicMin((unsigned long long*)&tbits[s], f64_bits(tv));
          cand[i].y = pos | NEAR_MARK;
        } else {
          atomicMin((int*)&best_pos[s], pos);
        }
      }
      if (i2 < n) {
        c2.x = checked(c2.x, n_samples, wcnt, 6);
        c2.y = checked(c2.y, n_group, wcnt, 7);
      }
      ti = checked(ti1_raw, n_tris, wcnt, 8);
      c = c1;
      c1 = c2;
      i += stride;
    }
  }
  wave_count(ap->cnt, C_EXACT, (unsigned long long)n_exact);
}

// k_search_exact when the rays with a single candidate are decided where they are shaded (SearchArgs::singles): one
// lane per candidate, and only the candidates of rays that have several (best_pos >= 0) are examined -- one in a few
// hundred on a closed mesh -- so the loop is kept plain.
__global__ void __launch_bounds__(256)
k_search_exact_multi(const SearchArgs* __restrict__ ap, int parity) {
  DevCounters* cnt = ap->cnt;
  unsigned int n_exact = 0;
  if (!cnt->overflow) {
    const int list = blockIdx.x & (NSLOT - 1), part = blockIdx.x / NSLOT, parts = gridDim.x / NSLOT;
    // the search listed these candidates apart (multi_flush); if that list did not fit, every candidate is looked at
    const bool listed = !cnt->multi_overflow;
    const int list_cap = (listed ? ap->multi_cap : ap->cand_cap) / NSLOT;
    const int n = min(listed ? cnt->multi_n[parity][list][0] : cnt->cand_n[parity][list][0], list_cap);
    const auto* cand = as_global(listed ? ap->multi : ap->cand) + (size_t)list * list_cap;
    auto* best_pos = as_global(ap->best_pos);
    const d3 o = rtx::ld3(ap->cam);
    for (int i = part * blockDim.x + threadIdx.x; i < n; i += parts * blockDim.x) {
      const int2 c = cand[i];
      const int s = checked(c.x, ap->n_samples, cnt, 38);
      // the ray's only candidate (a ray with several keeps a value >= 0 whatever the atomicMin below makes of it)
      if (load_relaxed(&best_pos[s]) < 0) continue;
      const int pos = checked(c.y, ap->n_group, cnt, 39);
      const int ti = checked(ap->group[pos], ap->n_tris, cnt, 40);
      const rtmi_triangle& tr = ap->tris[ti];
      const d3 d = rtx::mk(ap->dirs[(size_t)s * 3 + 0], ap->dirs[(size_t)s * 3 + 1], ap->dirs[(size_t)s * 3 + 2]);
      double tv;
      d3 P;
      n_exact += 1;
      if (rtx::tri_hit(rtx::ld3(tr.v[0]), rtx::ld3(tr.v[1]), rtx::ld3(tr.v[2]), ap->planes[ti], o, d, &tv, &P))
        atomicMin((int*)&best_pos[s], pos);
    }
  }
  wave_count(cnt, C_EXACT, (unsigned long long)n_exact);
}

// Nearest-hit policy, after k_search_exact: one lane per candidate; a marked candidate (a hit) whose parameter
// equals the ray's smallest one competes for the ray with its list position (the first in list order among
// equal parameters wins).  The parameter is evaluated again: the same two operations, the same value.
__global__ void __launch_bounds__(256)
k_search_nearest(const SearchArgs* __restrict__ ap, int parity) {
  const DevCounters* cnt = ap->cnt;
  if (cnt->overflow) return;
  const int list = blockIdx.x & (NSLOT - 1), part = blockIdx.x / NSLOT, parts = gridDim.x / NSLOT;
  const int list_cap = ap->cand_cap / NSLOT;
  const int n = min(cnt->cand_n[parity][list][0], list_cap);
  auto* cand = as_global(ap->cand) + (size_t)list * list_cap;
  const auto* dirs = as_global(ap->dirs);
  const auto* planes = as_global(ap->planes);
  const auto* group = as_global(ap->group);
  auto* best_pos = as_global(ap->best_pos);
  const auto* tbits = as_global(reinterpret_cast<const unsigned long long*>(ap->tval));
  const d3 o = rtx::ld3(ap->cam);
  for (int i = part * blockDim.x + threadIdx.x; i < n; i += parts * blockDim.x) {
    const int2 c = cand[i];
    if (!(c.y & NEAR_MARK)) continue;
    const int s = checked(c.x, ap->n_samples, ap->cnt, 21);
    const int pos = checked(c.y & ~NEAR_MARK, ap->n_group, ap->cnt, 22);
    cand[i].y = pos;
    const int ti = checked(group[pos], ap->n_tris, ap->cnt, 23);
    const auto* plp = planes + ti;
    TriPlane pl;
    pl.nx = plp->nx; pl.ny = plp->ny; pl.nz = plp->nz; pl.c = plp->c;
    const auto* dp = dirs + (size_t)s * 3;
    const double tv = rtx::tri_t(pl, o, rtx::mk(dp[0], dp[1], dp[2]));
    if (f64_bits(tv) == tbits[s]) atomicMin((int*)&best_pos[s], pos);
  }
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
