// kernels_scan.h -- part of rtmi.hip, which includes its parts in a fixed order (each relies on the ones before it).
// The list scan (RTMI_SEARCH_LIST_SCAN): k_search_primary.
// Not a stand-alone header.
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
  int nearest;       // hit policy: 1 = the hit with the smallest parameter wins (tree search only)
  int singles;       // tree search: a ray with exactly ONE candidate gets best_pos = ~position from k_search_tree; that
                     // pair is then decided in FP64 where the ray is shaded (k_shade_candidates), not by
                     // k_search_exact, which only serves the rays with several candidates
  int2* multi;       // singles: the candidates of the rays that have several, NSLOT lists of multi_cap / NSLOT (may be null)
  int multi_cap;
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
    // The batch's flag word travels with its operands (13th pair of the batch) rather than as a scalar of its
    // own: hipcc 7.2 lost the update of a separate `flag = next flag` on the path that leaves the rare branch, so
    // the batch after a closely examined one ran with a stale flag (found by tests/gpu_fuzz.py; the ISA showed it).
    constexpr int NCUR = CONE_FLAG / 2 + 1;
    static_assert(CONE_FLAG % 2 == 0 && NCUR * 2 <= CONE_DWORDS, "the flag is the first half of the pair after the operands");
    f2 cur[NCUR];
#pragma unroll
    for (int c = 0; c < NCUR; ++c) cur[c] = cone[c];
    for (int b = 0; b < nbatch; ++b) {
      // request the next batch (one spare, zero-filled batch is allocated past the end)
      f2 nxt[NCUR];
      const f2* np_ = cone + (size_t)(b + 1) * (CONE_DWORDS / 2);
#pragma unroll
      for (int c = 0; c < NCUR; ++c) nxt[c] = np_[c];
      const bool flagged = __builtin_bit_cast(int, cur[NCUR - 1].x) != 0;

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

      if (__builtin_expect(__any(mx >= 1.0f) || flagged, 0)) {
        // ---- rare path: look closer at the eight triangles of this batch, in list order ----
        const SearchArgs* a2 = ap;
        asm volatile("" : RTMI_SCALAR_REG(a2));  // read the argument block here, not ahead of the loop
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
      for (int c = 0; c < NCUR; ++c) cur[c] = nxt[c];
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
