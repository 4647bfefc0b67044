// kernels_fast.h -- part of rtmi.hip, which includes its parts in a fixed order (each relies on the ones before it).
// The frame of a scene whose top-level list holds nothing but bounding spheres without a material -- what loadObjFile
// builds (ViewController.swift:101-104) -- when the caller asks for pixels only (RTMI_FLAG_NO_RAY_STATISTICS):
//
//   k_block_cull_compact      the 8x8 pixel blocks some sample of which can come near a bounding sphere: the live blocks.
//                             Per-sample arrays exist for those only
//   k_search_tree_fast        per wave of 256 samples: FP32 directions from the sample tables, the cone tree walk;
//                             emits (sample, list position) candidates, marked when the FP32 edge test is certain
//   k_search_exact_multi_fast per candidate of a ray that has several: Triangle.intersections in FP64
//   k_shade_candidates_fast   per candidate: the ray's intersection is decided -- by the FP32 margins where they are
//                             certain, else by the reference's FP64 arithmetic (sample direction, Sphere.intersections of
//                             the bounding sphere, Triangle.intersections) -- and shaded in FP32 with an error bound
//                             (rtmi_shade32.h)
//   k_resolve_fast            per pixel: sums colours and bounds; writes the bytes the bounds settle, lists the other pixels
//   k_fix_pixels              per listed pixel: every sample again with the reference's FP64 arithmetic
//
// There is no ray-generation launch, no per-sample direction array and no queue of samples: a ray that ends up
// without a triangle is black whether or not it meets the bounding sphere (RayTracingFunctions.swift:86-93), so the
// sphere's own test is only ever needed for the rays that have a candidate.  Hit ids and bytes are the reference's,
// like the general path's (the whole-frame digests hold both to the oracle's).
// Not a stand-alone header.

// The reference's FP64 direction of sample s (WorldCoordinateSequence.swift:62-78, RayTracingFunctions.swift:73).
__device__ __forceinline__ d3 fast_sample_dir64(const FrameConst& fc, const int2* __restrict__ blocks, int s) {
  int x, y, kx, ky;
  fast_sample_coords(fc, blocks, s, &x, &y, &kx, &ky);
  return sample_dir(fc, x, y, kx, ky);
}

// raytracePixels takes intersections[0] of the top-level list (RTF:76-85): the first member, in list order, that the ray
// intersects at all.  Every member is a bounding sphere here.
__device__ __forceinline__ int fast_first_object(const SceneDev& sc, d3 o, d3 d) {
  for (int k = 0; k < sc.n_objects; ++k)
    if (rtx::sphere_any_ahead(rtx::ld3(sc.objects[k].center), sc.objects[k].radius, o, d)) return k;
  return -1;
}

// k_search_exact_multi with the direction taken from the sample tables.
__global__ void __launch_bounds__(256)
k_search_exact_multi_fast(const SearchArgs* __restrict__ ap, int parity, FrameConst fc, FastArgs fa) {
  DevCounters* cnt = ap->cnt;
  fast_live_counts(fa, fc, cnt);
  unsigned int n_exact = 0;
  if (!cnt->overflow) {
    const int list = blockIdx.x & (NSLOT - 1), part = blockIdx.x / NSLOT, parts = gridDim.x / NSLOT;
    const bool listed = !cnt->multi_overflow;  // as in k_search_exact_multi
    const int list_cap = (listed ? ap->multi_cap : ap->cand_cap) / NSLOT;
    const int n = min(listed ? cnt->multi_n[parity][list][0] : cnt->cand_n[parity][list][0], list_cap);
    const auto* cand = as_global(listed ? ap->multi : ap->cand) + (size_t)list * list_cap;
    auto* best_pos = as_global(ap->best_pos);
    const d3 o = rtx::ld3(ap->cam);
    for (int i = part * blockDim.x + threadIdx.x; i < n; i += parts * blockDim.x) {
      const int2 c = cand[i];
      const int s = checked(c.x, fa.nsamples, cnt, 41);
      if (load_relaxed(&best_pos[s]) < 0) continue;  // the ray's only candidate: decided where it is shaded
      const int pos = checked(c.y & ~SURE_BIT, ap->n_group, cnt, 42);
      const int ti = checked(ap->group[pos], ap->n_tris, cnt, 43);
      const rtmi_triangle& tr = ap->tris[ti];
      const d3 d = fast_sample_dir64(fc, fa.blocks, s);
      double tv;
      d3 P;
      n_exact += 1;
      if (rtx::tri_hit(rtx::ld3(tr.v[0]), rtx::ld3(tr.v[1]), rtx::ld3(tr.v[2]), ap->planes[ti], o, d, &tv, &P))
        atomicMin((int*)&best_pos[s], pos);
    }
  }
  wave_count(cnt, C_EXACT, (unsigned long long)n_exact);
}

#ifndef RTMI_SHADE_FAST_WAVES
#define RTMI_SHADE_FAST_WAVES 5
#endif
constexpr int FAST_LDS_LIGHTS = 8;  // a frame of this path has at most so many lights (the host sends others down the general path)
constexpr int FAST_LDS_MATS = 8;    // materials kept in LDS; the others are read from memory
// One lane per candidate (grid-stride over the NSLOT lists, like k_shade_candidates).  A ray's only candidate
// (best_pos < 0) is decided here; of a ray with several, the candidate whose position equals best_pos is the triangle
// k_search_exact_multi_fast settled on.  Either way the ray must also have the group's bounding sphere as the first
// member of the list it intersects.  Both questions are answered by FP32 margins when those are certain -- the
// candidate's mark (inside all three edge planes by the filter's margin), n.d clear of zero, the sphere's a and delta
// clear of their error -- and by the reference's FP64 arithmetic otherwise (about one candidate in a few thousand, but
// every candidate of a scene with several bounding spheres that is not the first one's).
// The per-frame constants of the lighting sit in LDS (a read from there cannot be mistaken for one that the kernel's
// own stores might change, so the compiler is free to schedule it early); a step's loads -- the sample's block, then
// the table entries, the plane, the material index -- depend only on the candidate entry, which is fetched one step ahead.
__global__ void __launch_bounds__(256, RTMI_SHADE_FAST_WAVES)
k_shade_candidates_fast(FrameConst fc, SceneDev sc, FastArgs fa, const SearchArgs* __restrict__ ap, int par) {
  __shared__ rtx32::LightF s_lights[FAST_LDS_LIGHTS];
  __shared__ rtx32::MatF s_mats[FAST_LDS_MATS];
  {
    const int nl = min(sc.n_lights, FAST_LDS_LIGHTS), nm = min(fa.n_materials, FAST_LDS_MATS);
    constexpr int LW = sizeof(rtx32::LightF) / 4, MW = sizeof(rtx32::MatF) / 4;
    for (int k = threadIdx.x; k < nl * LW; k += blockDim.x) reinterpret_cast<int*>(s_lights)[k] = reinterpret_cast<const int*>(fa.lights)[k];
    for (int k = threadIdx.x; k < nm * MW; k += blockDim.x) reinterpret_cast<int*>(s_mats)[k] = reinterpret_cast<const int*>(fa.mats)[k];
  }
  __syncthreads();
  DevCounters* cnt = (DevCounters*)as_global(ap->cnt);
  fast_live_counts(fa, fc, cnt);
  unsigned int n_hit = 0, n_exact = 0, n_sure = 0;
  if (!cnt->overflow) {
    const int list = blockIdx.x & (NSLOT - 1), part = blockIdx.x / NSLOT, parts = gridDim.x / NSLOT;
    const int list_cap = ap->cand_cap / NSLOT;
    const int n = min(cnt->cand_n[par][list][0], list_cap);
    const auto* cand = as_global(ap->cand) + (size_t)list * list_cap;
    const int stride = parts * blockDim.x;
    const auto* best_pos = as_global(ap->best_pos);
    const auto* blocks = as_global(fa.blocks);
    auto* hit_id = as_global(fa.hit_id);
    auto* rgbf = as_global(fa.rgbf);
    const auto* pos_plane = as_global(fa.pos_plane);
    const auto* pos_mt = as_global(fa.pos_mt);
    const auto* utab_f = as_global(fc.utab_f);
    const auto* vtab_f = as_global(fc.vtab_f);
    const int n_group = ap->n_group, n_tris = ap->n_tris;
    const int n_blocks = max(fa.nsamples / (BLK_PIX * fc.spp), 1);
    const SphereF sph = fa.spheres[fa.obj];
    // what a step needs before anything else: the entry (both indices range-checked, the mark stays on the position), the
    // ray's best position, the sample's block and sub-sample
    struct Head { int s, posm, bp, kx, ky; int2 bc; };
    auto head = [&](int k) {
      Head h;
      const int2 e = cand[k];
      h.s = checked(e.x, fa.nsamples, cnt, 44);
      h.posm = checked(e.y & ~SURE_BIT, n_group, cnt, 45) | (e.y & SURE_BIT);
      h.bp = best_pos[h.s];
      int b;
      group_coords(fc, h.s >> 6, &b, &h.kx, &h.ky);
      h.bc = blocks[min(b, n_blocks - 1)];
      return h;
    };
    int i = part * blockDim.x + threadIdx.x;
    Head h{};
    if (i < n) h = head(i);
    for (int base = part * blockDim.x; base < n; base += stride) {
      const int inext = i + stride;
      Head hn{};
      if (inext < n) hn = head(inext);
      const int s = h.s, pos = h.posm & ~SURE_BIT;
      const bool single = h.bp < 0;
      if (i < n && (single || h.bp == pos)) {
        const int pix = s & (BLK_PIX - 1);
        const int x = h.bc.x * BLK + (pix & (BLK - 1)), y = h.bc.y * BLK + (pix >> 3);
        // this step's loads, none depending on another
        const float* uc = utab_f + (size_t)(x * fc.n1 + h.kx) * 3;
        const float* vc = vtab_f + (size_t)(y * fc.n1 + h.ky) * 3;
        const float u0 = uc[0], u1 = uc[1], u2 = uc[2], v0 = vc[0], v1 = vc[1], v2 = vc[2];
        const float4 pl = pos_plane[pos];
        const int2 mt = pos_mt[pos];  // material index, triangle index
        // the FP32 direction, as the search derived it (fast_dir)
        const float fx = fc.base_f[0] + u0 - v0, fy = fc.base_f[1] + u1 - v1, fz = fc.base_f[2] + u2 - v2;
        const float inv = rtx32::rsq(fx * fx + fy * fy + fz * fz);
        const float dx = fx * inv, dy = fy * inv, dz = fz * inv;
        const int ti = checked(mt.y, n_tris, cnt, 46);
        // certain by FP32?  the triangle: marked by the search, and n.d clear of zero (Renderables.swift:146)
        const float nd = -(pl.x * dx + pl.y * dy + pl.z * dz);
        const bool sure_tri = !single || ((h.posm & SURE_BIT) && nd > 1.0e-4f);
        // the sphere: a = -d.cte clear of 1e-7 (:274) and delta = a^2 - (|cte|^2 - r^2) clear of zero (:255)
        const float a = -(dx * sph.cx + dy * sph.cy + dz * sph.cz);
        const float delta = a * a - sph.k;
        const float ed = 2.0f * fabsf(a) * sph.ea + sph.ea * sph.ea + 8.0f * rtx32::U * (a * a + sph.kabs);
        const bool sure_sph = fa.obj == 0 && a > 1.0e-6f + 2.0f * sph.ea && delta > 2.0f * ed;
        bool hit = sure_tri && sure_sph && !(fa.force & 1);
        if (!hit) {  // the reference's arithmetic
          const d3 o = rtx::ld3(fc.cam);
          const d3 d = sample_dir(fc, x, y, h.kx, h.ky);
          hit = fast_first_object(sc, o, d) == fa.obj;
          if (hit && single) {
            const rtmi_triangle& tr = sc.tris[ti];
            double tv;
            d3 P;
            hit = rtx::tri_hit(rtx::ld3(tr.v[0]), rtx::ld3(tr.v[1]), rtx::ld3(tr.v[2]), sc.planes[ti], o, d, &tv, &P);
            n_exact += 1;
          }
        } else if (single) {
          n_sure += 1;
        }
        if (hit) {
          n_hit += 1;
          hit_id[s] = ti;
          const int mi = mt.x;
          rtx32::Shade32 col;
          col.r = col.g = col.b = col.err = 0.0f;  // no material: RGB.black (RTF:192-196, :245)
          if (mi >= 0) {
            rtx32::MatF m;
            if (mi < FAST_LDS_MATS) m = s_mats[mi];
            else m = fa.mats[min(mi, fa.n_materials - 1)];
            // (ns_uniform is a kernel argument: the branch and the powering loop behind it are scalar code)
            if (fa.ns_uniform) col = rtx32::shade_hit<true>(dx, dy, dz, fa.dir_err, pl.x, pl.y, pl.z, pl.w, m, s_lights, sc.n_lights, fa.ns_uniform);
            else col = rtx32::shade_hit<false>(dx, dy, dz, fa.dir_err, pl.x, pl.y, pl.z, pl.w, m, s_lights, sc.n_lights);
          }
          rgbf[s] = make_float4(col.r, col.g, col.b, col.err);
        }
      }
      h = hn;
      i = inext;
    }
  }
  search_lists_retire(cnt, par);
  wave_count(cnt, C_HITS, n_hit);
  wave_count(cnt, C_EXACT, n_exact);
  wave_count(cnt, C_FP32_DECIDED, n_sure);
}

// Per pixel: RGB.average and the x255 truncation (Utilities.swift:54-65, RayTracingFunctions.swift:105-111) from the
// FP32 colours and their bounds.  The sums are formed in FP64 (exact for float addends); the reference's byte is
// floor(255 * mean) of a mean that lies within the summed bounds of ours, so when both ends of that interval truncate
// to the same byte that byte is the reference's.  Otherwise -- a sample was flagged, or the interval holds a whole
// number -- the pixel goes to one of NSLOT lists for k_fix_pixels (workgroup-wide append, one atomic per workgroup and
// list) and nothing is written for it here.
__global__ void __launch_bounds__(256)
k_resolve_fast(FrameConst fc, FastArgs fa, int npix, DevCounters* cnt, unsigned char* rgba, int* hit_out) {
  __shared__ int wtot[4];
  __shared__ int gbase;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  bool fix = false;
  int x = 0, y = 0;
  const int b = p >> 6, pix = p & (BLK_PIX - 1);  // b counts the batch's blocks as the host listed them
  if (p < npix && pixel_coords(fc, fa.all_blocks, b, pix, &x, &y)) {
    const size_t first_byte = (size_t)4 * fc.W * y + (size_t)x * 4;  // RayTracingFunctions.swift:106-107
    const int slot = fa.block_slot[b];  // among the live blocks: the per-sample arrays are indexed by it
    if (slot < 0) {
      // no sample of this block can hit anything (k_block_cull): every sample is RGB.black
      *reinterpret_cast<unsigned int*>(rgba + first_byte) = 255u << 24;
      if (hit_out)
        for (int q = 0; q < fc.spp; ++q) hit_out[((size_t)y * fc.W + x) * fc.spp + q] = -1;
    } else {
      double sr = 0.0, sg = 0.0, sb = 0.0, er = 0.0, eg = 0.0, eb = 0.0;
      for (int q = 0; q < fc.spp; ++q) {
        const int s = ((slot * fc.spp + q) << 6) + pix;
        const int id = fa.hit_id[s];
        if (id >= 0) {
          const float4 v = fa.rgbf[s];
          sr += (double)v.x; sg += (double)v.y; sb += (double)v.z;
          // a channel that holds exactly 1 is exactly 1 in the reference (rtx32::Shade32)
          er += v.x == 1.0f ? 0.0 : (double)v.w;
          eg += v.y == 1.0f ? 0.0 : (double)v.w;
          eb += v.z == 1.0f ? 0.0 : (double)v.w;
        }
        if (hit_out) hit_out[((size_t)y * fc.W + x) * fc.spp + q] = id;
      }
      const double factor = 255.0 / (double)fc.spp;
      auto settle = [&](double sum, double e, unsigned int* byte) {
        if (e == 0.0) {
          // every addend is the reference's own value (exactly 0 or exactly 1), added in the reference's order: the
          // reference's operations on it give the reference's byte (Utilities.swift:61-62, RTF:108-110)
          const double f1 = 1 / (double)fc.spp;
          *byte = (unsigned int)(unsigned char)(255 * (f1 * sum));
          return true;
        }
        // the reference's own roundings (sum, 1/count, two products) are below 1e-12 relative
        double lo = factor * (sum - e) * (1.0 - 1.0e-12) - 1.0e-9, hi = factor * (sum + e) * (1.0 + 1.0e-12) + 1.0e-9;
        lo = fmax(lo, 0.0);    // a clamped colour is never negative
        hi = fmin(hi, 255.0);  // nor above 1
        const double fl = floor(lo);
        *byte = (unsigned int)fl;
        return lo <= hi && fl == floor(hi);  // false for NaN and for an infinite bound
      };
      unsigned int br, bg, bb;
      const bool okr = settle(sr, er, &br), okg = settle(sg, eg, &bg), okb = settle(sb, eb, &bb);
      const bool ok = okr && okg && okb && !(fa.force & 2);
      if (ok) *reinterpret_cast<unsigned int*>(rgba + first_byte) = br | (bg << 8) | (bb << 16) | (255u << 24);
      else fix = true;
    }
  }
  // append the pixels to fix: list = workgroup & 63, one returning atomic per workgroup
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const unsigned long long mask = __ballot(fix);
  if (lane == 0) wtot[w] = __popcll(mask);
  __syncthreads();
  const int list = blockIdx.x & (NSLOT - 1);
  if (threadIdx.x == 0) {
    const int tot = wtot[0] + wtot[1] + wtot[2] + wtot[3];
    gbase = tot ? atomicAdd(&cnt->fix_n[list][0], tot) : 0;
  }
  __syncthreads();
  if (fix) {
    int at = gbase + __popcll(mask & ((1ull << lane) - 1ull));
    for (int q = 0; q < w; ++q) at += wtot[q];
    const int list_cap = fa.fix_cap / NSLOT;
    if (at < list_cap) fa.fix_list[(size_t)list * list_cap + at] = p;
    else cnt->bug = 47;  // cannot happen: a list holds every pixel of the workgroups that append to it
  }
}

// One WAVE per listed pixel: its samples, 64 at a time, are evaluated with the reference's FP64 arithmetic -- the
// direction from the tables, the triangle's parameter, calculateLighting (the instantiation without secondary rays)
// -- then lane 0 adds the colours in sample order as RGB.average does and stores the pixel.
__global__ void __launch_bounds__(256)
k_fix_pixels(FrameConst fc, ShadeConst shc, SceneDev sc, FastArgs fa, int npix, DevCounters* cnt, unsigned char* rgba) {
  __shared__ double col[4][64][3];
  fast_live_counts(fa, fc, cnt);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int list = blockIdx.x & (NSLOT - 1), part = blockIdx.x / NSLOT, parts = gridDim.x / NSLOT;
  const int list_cap = fa.fix_cap / NSLOT;
  const int n = min(cnt->fix_n[list][0], list_cap);
  const int* fix = fa.fix_list + (size_t)list * list_cap;
  const TopTreeRef tt{};  // no secondary ray is traced here
  unsigned int n_fixed = 0;
  for (int i = part * 4 + w; i < n; i += parts * 4) {  // wave-uniform
    const int p = fix[i];
    const int b = p >> 6, pix = p & (BLK_PIX - 1);
    int x, y;
    int slot = -1;
    if ((unsigned)p < (unsigned)npix) slot = fa.block_slot[b];
    if (slot < 0 || slot >= fa.cap_live || !pixel_coords(fc, fa.all_blocks, b, pix, &x, &y)) {
      cnt->bug = 48;
      continue;
    }
    double cr = 0.0, cg = 0.0, cb = 0.0;  // cumulative = RGB(0,0,0), Utilities.swift:56
    for (int q0 = 0; q0 < fc.spp; q0 += 64) {
      const int q = q0 + lane;
      RGBd c = {0.0, 0.0, 0.0};  // RGB.black for a miss
      if (q < fc.spp) {
        const int s = ((slot * fc.spp + q) << 6) + pix;
        int id = fa.hit_id[s];
        if (id >= 0) {
          id = checked(id, sc.n_tris, cnt, 49);
          const d3 cam = rtx::ld3(fc.cam);
          const d3 d = fast_sample_dir64(fc, fa.blocks, s);
          const TriPlane pl = sc.planes[id];
          Hit h;
          h.t = rtx::tri_t(pl, cam, d);            // the pair was accepted: the same two operations as Triangle.intersections
          h.P = rtx::add(cam, rtx::smul(h.t, d));  // Renderables.swift:153
          h.n = rtx::mk(pl.nx, pl.ny, pl.nz);
          h.has_normal = 1; h.n_unit = 1; h.material = sc.tris[id].material; h.id = id;
          tally_t t0 = 0, t1 = 0, t2 = 0, t3 = 0;
          bool defer = false;
          c = calculate_lighting<true, false, true, true, false>(sc, tt, shc, h, cam, &t0, &t1, &t2, &t3, &defer);
        }
      }
      col[w][lane][0] = c.r; col[w][lane][1] = c.g; col[w][lane][2] = c.b;
      __builtin_amdgcn_wave_barrier();
      if (lane == 0) {
        const int m = min(64, fc.spp - q0);
        for (int k = 0; k < m; ++k) { cr = cr + col[w][k][0]; cg = cg + col[w][k][1]; cb = cb + col[w][k][2]; }
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) {
      const double factor = 1 / (double)fc.spp;  // Utilities.swift:61-62
      const double ar = factor * cr, ag = factor * cg, ab = factor * cb;
      const unsigned int px = (unsigned int)(unsigned char)(255 * ar) | ((unsigned int)(unsigned char)(255 * ag) << 8) |
                              ((unsigned int)(unsigned char)(255 * ab) << 16) | (255u << 24);
      *reinterpret_cast<unsigned int*>(rgba + (size_t)4 * fc.W * y + (size_t)x * 4) = px;
      n_fixed += 1;
    }
  }
  wave_count(cnt, C_FIX_PIXELS, n_fixed);
}
