// kernels_secondary.h -- part of rtmi.hip, which includes its parts in a fixed order (each relies on the ones before it).
// Secondary rays: top-level tests in list order and the bounding-sphere tree walk.
// Not a stand-alone header.
// ---------------------------------------------------------------------------------------
// top-level tests against spheres, bounding spheres and lights (FP64, list order)
// ---------------------------------------------------------------------------------------
// Per-lane work tallies of the shading kernels (rays, pairs); a lane serves a few hundred samples of a batch, so
// 32 bits hold them, and they are widened when a wave adds its sum to the 64-bit counters.
typedef unsigned int tally_t;

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
      if (rtx::sphere_any_ahead(rtx::ld3(ob.center), ob.radius, o, d)) return k;
    } else if (ob.kind == RTMI_LIGHT) {
      double t;
      d3 P;
      if (rtx::light_hit(rtx::ld3(sc.lights[ob.first].loc), o, d, &t, &P)) return k;
    }
  }
  return -1;
}

// Nearest-hit policy: the sphere-like member whose first intersection has the smallest parameter (the first in
// list order among equal parameters); -1 if none.  A sphere's own roots are sorted near first (REN:265-271).
__device__ inline int nearest_sphere_like(const SceneDev& sc, d3 o, d3 d, double* t_out) {
  int best = -1;
  double best_t = 0.0;
  for (int k = 0; k < sc.n_objects; ++k) {
    const rtmi_object& ob = sc.objects[k];
    double t;
    d3 P, n;
    bool hit = false;
    if (ob.kind == RTMI_SPHERE || ob.kind == RTMI_BOUNDING_SPHERE) hit = rtx::sphere_first_raw(rtx::ld3(ob.center), ob.radius, o, d, &t, &P, &n);
    else if (ob.kind == RTMI_LIGHT) hit = rtx::light_hit(rtx::ld3(sc.lights[ob.first].loc), o, d, &t, &P);
    if (hit && (best < 0 || t < best_t)) { best = k; best_t = t; }
  }
  *t_out = best_t;
  return best;
}

// The bounding-sphere tree of the top-level triangle runs as the shading kernels see it.  The
// three arrays are separate __restrict__ kernel arguments (that is what lets the compiler fetch
// records through scalar loads in kernels that also store).
struct TopTreeRef {
  const f2* rec;        // sphere records, all levels (tree_sphere_records)
  const float* gst2;    // per top-level ordinal: plane and edge operands (k_pack_general)
  const int* slot_pos;  // slot -> top-level ordinal (= position among the top-level triangles)
  int ltop, n_root, n_slots, n_records;
  int cone_levels;      // bit l: the records of level l have cones of normals (a second block after the n_records sphere records)
  int off[TREE_MAX_LEVELS];
};

// The scalar part of TopTreeRef, passed to the kernels by value.
struct TopTreeArgs {
  int ltop, n_root, n_slots, n_records;
  int cone_levels;
  int off[TREE_MAX_LEVELS];
};

__device__ inline TopTreeRef make_top_tree(const TopTreeArgs& a, const f2* rec, const float* gst2, const int* slot_pos) {
  TopTreeRef t;
  t.rec = rec; t.gst2 = gst2; t.slot_pos = slot_pos;
  t.ltop = a.ltop; t.n_root = a.n_root; t.n_slots = a.n_slots; t.n_records = a.n_records;
  t.cone_levels = a.cone_levels;
#pragma unroll
  for (int l = 0; l < TREE_MAX_LEVELS; ++l) t.off[l] = a.off[l];
  return t;
}

// The shadow-ray extensions (RTMI_FLAG_SHADOW_MAX_DISTANCE, RTMI_FLAG_SHADOWS_ENTER_BOUNDED) walk trees of the same
// layout over other member lists, with more conditions on a hit.  members/n_members/c0 replace
// SceneDev::top_group/n_top_tris/c0; a hit counts only below max_t; with member_owner (member -> the bounding
// sphere that lists it) a hit counts only if the ray intersects that sphere at all, which is when the
// extension tests the sphere's children.
struct WalkExt {
  const int* members;       // member ordinal -> triangle index
  const int* member_owner;  // member ordinal -> object index of its bounding sphere, or nullptr
  long long n_members;
  double c0[3];
  double max_t;
  int only_owner;           // >= 0 (with member_owner): only the members of this bounding sphere count, and the ray is
                            // known to intersect it (RTMI_FLAG_REFLECTIONS_ENTER_BOUNDED)
};

// PRECONDITION of the three walks below (top_tree_trace, top_tree_trace_nearest, and shadow_blocked_ext /
// reflection_enters_bounded through them): |d| = 1 to within FP64 rounding.  The half-line test takes A = d.(c - o) as a
// distance along the ray and the cone-of-normals test compares a.d with sin(theta): a shorter d would reject spheres the
// ray passes, a longer one would pass over front-facing nodes.  Every caller passes the output of rtx::normalize or
// the reflection of one unit vector about another (RTF:209-211, :227), which is what the reference traces with too:
// its Sphere.intersections (REN:251-253) assumes the same.  The CPU build of the kernels checks it.
__device__ inline void walk_wants_unit_direction(d3 d) {
#if !defined(__HIPCC__)  // the CPU build of the kernels (tests/emu)
  const double l2 = d.x * d.x + d.y * d.y + d.z * d.z;
  if (l2 == l2 && !(fabs(l2 - 1.0) < 1e-9)) {  // (a NaN direction hits nothing either way)
    fprintf(stderr, "secondary-ray walk called with |d|^2 = %.17g\n", l2);
    abort();
  }
#endif
}

// A general-origin ray (reflection or shadow ray) against ALL top-level triangles: the ordinal of
// the hit that comes first in list order (any_only: of some hit), or -1.  The lanes that call this
// together walk the tree together, like k_search_tree: a record of eight bounding spheres is
// evaluated for every calling lane's own ray (packed half-line-sphere test, two spheres per
// instruction), and a child is entered if some lane passes.  At a leaf the per-triangle cascade
// of k_pack_general follows, then rtx::tri_hit in FP64.  Called from divergent shading code: the
// wave-level votes only see the lanes that are tracing.
// EXT: the tree and the conditions of *WX (see WalkExt); any_only walks only.
template <bool EXT = false>
__device__ inline int top_tree_trace(const SceneDev& sc, const TopTreeRef& T, d3 o, d3 d, bool any_only, double* t_out,
                                     d3* P_out, tally_t* n_tests, const WalkExt* WX = nullptr) {
  walk_wants_unit_direction(d);
  const double c0x = EXT ? WX->c0[0] : sc.c0[0], c0y = EXT ? WX->c0[1] : sc.c0[1], c0z = EXT ? WX->c0[2] : sc.c0[2];
  const long long n_members = EXT ? WX->n_members : sc.n_top_tris;
  const int* members = EXT ? WX->members : sc.top_group;
  const float ox = (float)(o.x - c0x), oy = (float)(o.y - c0y), oz = (float)(o.z - c0z);
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
    // A is the parameter of the line's closest approach to the centre.  A triangle is only hit at a positive
    // parameter (REN:152), and when the closest approach lies behind the origin every point ahead is farther from
    // the centre than the origin is: then the sphere can only hold a hit if it holds the origin, B + nk >= 0.
    // (A wrongly rounded to below zero is smaller than the margin in nk squared away.)
    A.x = __builtin_fmaxf(A.x, 0.0f);
    A.y = __builtin_fmaxf(A.y, 0.0f);
    return __builtin_elementwise_fma(A, A, B) + nk;
  };
  auto level_off = [&](int lvl) {
    return lvl == 1 ? T.off[1] : lvl == 2 ? T.off[2] : lvl == 3 ? T.off[3] : lvl == 4 ? T.off[4] : lvl == 5 ? T.off[5]
         : lvl == 6 ? T.off[6] : T.off[7];
  };
  auto eval = [&](int lvl, int ri) -> unsigned {
    const f2* rp = T.rec + (size_t)min(level_off(lvl) + ri, T.n_records - 1) * (REC_DWORDS / 2);  // the clamp never binds
    const unsigned long long live = wave_mask(!any_only || best == 0x7fffffff);
    unsigned mask = 0;
    if ((T.cone_levels >> lvl) & 1) {
      // a node whose triangles all face away from the ray is passed over (tree_sphere_records)
      const f2* cp = rp + (size_t)T.n_records * (REC_DWORDS / 2);
#pragma unroll
      for (int pr = 0; pr < 4; ++pr) {
        const f2 U = pair_u(rp, pr);
        const f2 ax = cp[pr * 4 + 0], ay = cp[pr * 4 + 1], az = cp[pr * 4 + 2], thr = cp[pr * 4 + 3];
        const f2 C = __builtin_elementwise_fma(az, Z, __builtin_elementwise_fma(ay, Y, ax * X));
        if (live & wave_mask(!(U.x < 0.0f)) & wave_mask(C.x < thr.x)) mask |= 1u << (2 * pr);
        if (live & wave_mask(!(U.y < 0.0f)) & wave_mask(C.y < thr.y)) mask |= 2u << (2 * pr);
      }
    } else {
#pragma unroll
      for (int pr = 0; pr < 4; ++pr) {
        f2 U = pair_u(rp, pr);
        if (live & wave_mask(!(U.x < 0.0f))) mask |= 1u << (2 * pr);
        if (live & wave_mask(!(U.y < 0.0f))) mask |= 2u << (2 * pr);
      }
    }
    return mask;
  };
  auto leaf = [&](int ri) {
    if ((unsigned)(ri * 8) >= (unsigned)T.n_slots) return;  // cannot happen (absent nodes never pass); keeps a bad record from becoming a bad address
    const f2* rp = T.rec + (size_t)(T.off[0] + ri) * (REC_DWORDS / 2);
    const bool live = !any_only || best == 0x7fffffff;
    const unsigned long long live_m = wave_mask(live);
    if (live) ++leaf_visits;
#pragma unroll
    for (int pr = 0; pr < 4; ++pr) {
      f2 U = pair_u(rp, pr);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float u = h ? U.y : U.x;
        // `cand` is the lane's own answer, `cm` the same for the whole wave (see wave_mask)
        bool cand = live && !(u < 0.0f);
        unsigned long long cm = live_m & wave_mask(!(u < 0.0f));
        if (!cm) continue;
        const int g = T.slot_pos[ri * 8 + pr * 2 + h];  // top-level ordinal
        if ((unsigned)g >= (unsigned)n_members) continue;  // padding slot (its sphere passes nothing; belt and braces)
        // a later list position cannot replace the hit
        cand = cand && (any_only ? best == 0x7fffffff : g < best);
        cm &= any_only ? wave_mask(best == 0x7fffffff) : wave_mask(g < best);
        if (!cm) continue;
        const float* s2 = T.gst2 + (size_t)g * GST2_FLOATS;
        {
          const float nx = s2[0], ny = s2[1], nz = s2[2], cpl = s2[3], Ea = s2[4], Eb = s2[5];
          float nd = __builtin_fmaf(nz, dz, __builtin_fmaf(ny, dy, nx * dx));
          float no = __builtin_fmaf(nz, oz, __builtin_fmaf(ny, oy, nx * ox)) - cpl;
          cand = cand && (nd <= Ea) && (no >= -Eb);
          cm &= wave_mask(nd <= Ea) & wave_mask(no >= -Eb);
        }
        if (!cm) continue;
#pragma unroll
        for (int e = 0; e < 3; ++e) {
          const float* w = s2 + 9 + e * 6;
          float sv = __builtin_fmaf(w[2], dz, __builtin_fmaf(w[1], dy, w[0] * dx));
          sv = __builtin_fmaf(w[5], mz_, __builtin_fmaf(w[4], my_, __builtin_fmaf(w[3], mx_, sv)));
          cand = cand && (sv <= s2[6 + e]);
          cm &= wave_mask(sv <= s2[6 + e]);
        }
        if (!cm) continue;
        if (cand) {
          const int ti = min(max(members[g], 0), sc.n_tris - 1);  // the clamps never bind
          const rtmi_triangle& tr = sc.tris[ti];
          double t;
          d3 P;
          bool mine = true;
          if constexpr (EXT) mine = !(WX->member_owner && WX->only_owner >= 0) || WX->member_owner[g] == WX->only_owner;
          bool hit = mine && rtx::tri_hit(rtx::ld3(tr.v[0]), rtx::ld3(tr.v[1]), rtx::ld3(tr.v[2]), sc.planes[ti], o, d, &t, &P);
          if constexpr (EXT) {
            hit = hit && t < WX->max_t;
            if (hit && WX->member_owner && WX->only_owner < 0) {
              const rtmi_object& ob = sc.objects[min(max(WX->member_owner[g], 0), sc.n_objects - 1)];
              hit = rtx::sphere_any(rtx::ld3(ob.center), ob.radius, o, d);
            }
          }
          if (hit) {
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
  *n_tests += 8u * leaf_visits;  // (ray, triangle) sphere tests this lane's ray took part in
  return best == 0x7fffffff ? -1 : best;
}

// The sphere-like members of the top-level list (spheres, bounding spheres, lights) for a secondary ray, in list
// order: intersections[0] among them if want_first (*first_obj = its list index), and whether any of them other
// than a bounding sphere is hit.  For a shadow ray the first such hit settles the question.
__device__ inline bool secondary_objects(const SceneDev& sc, d3 o, d3 d, bool want_first, Hit* h, bool* any_non_bounding,
                                         int* first_obj_out) {
  bool found = false, anb = false;
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
  *any_non_bounding = anb;
  *first_obj_out = first_obj;
  return found;
}

// traceRay over the whole top-level list (RayTracingFunctions.swift:37-44) for a secondary
// ray: intersections[0] if want_first, and whether any non-bounding object is hit.
template <bool HAS_TOP>
__device__ inline bool trace_top_secondary(const SceneDev& sc, const TopTreeRef& tt, d3 o, d3 d, bool want_first, Hit* h,
                                           bool* any_non_bounding, tally_t* n_tests) {
  // spheres, bounding spheres and lights first, in list order: for a shadow ray the first
  // non-bounding hit settles the question and the triangles need not be searched
  bool anb = false;
  int first_obj = 0x7fffffff;
  bool found = secondary_objects(sc, o, d, want_first, h, &anb, &first_obj);
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

// Nearest-hit policy (an extension; the reference's list-order functions above are left as they are): a copy of
// top_tree_trace in which the hit with the smallest parameter wins, the lower ordinal among equal parameters.
// A general-origin ray (reflection or shadow ray) against ALL top-level triangles: the ordinal of
// the hit that comes first in list order (any_only: of some hit), or -1.  The lanes that call this
// together walk the tree together, like k_search_tree: a record of eight bounding spheres is
// evaluated for every calling lane's own ray (packed half-line-sphere test, two spheres per
// instruction), and a child is entered if some lane passes.  At a leaf the per-triangle cascade
// of k_pack_general follows, then rtx::tri_hit in FP64.  Called from divergent shading code: the
// wave-level votes only see the lanes that are tracing.
// EXT: the tree and the member filter of *WX (see WalkExt; only_owner walks).
template <bool EXT = false>
__device__ inline int top_tree_trace_nearest(const SceneDev& sc, const TopTreeRef& T, d3 o, d3 d, bool any_only, double* t_out,
                                     d3* P_out, tally_t* n_tests, const WalkExt* WX = nullptr) {
  walk_wants_unit_direction(d);
  const bool nearest = !any_only;  // the hit with the smallest parameter instead of the lowest ordinal
  double best_t = 0.0;
  const double c0x = EXT ? WX->c0[0] : sc.c0[0], c0y = EXT ? WX->c0[1] : sc.c0[1], c0z = EXT ? WX->c0[2] : sc.c0[2];
  const long long n_members = EXT ? WX->n_members : sc.n_top_tris;
  const int* members = EXT ? WX->members : sc.top_group;
  const float ox = (float)(o.x - c0x), oy = (float)(o.y - c0y), oz = (float)(o.z - c0z);
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
    // A is the parameter of the line's closest approach to the centre.  A triangle is only hit at a positive
    // parameter (REN:152), and when the closest approach lies behind the origin every point ahead is farther from
    // the centre than the origin is: then the sphere can only hold a hit if it holds the origin, B + nk >= 0.
    // (A wrongly rounded to below zero is smaller than the margin in nk squared away.)
    A.x = __builtin_fmaxf(A.x, 0.0f);
    A.y = __builtin_fmaxf(A.y, 0.0f);
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
        if ((unsigned)g >= (unsigned)n_members) continue;  // padding slot (its sphere passes nothing; belt and braces)
        // a later list position cannot replace the hit (under the nearest-hit policy any triangle can)
        cand = cand && (any_only ? best == 0x7fffffff : (nearest || g < best));
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
          const int ti = min(max(members[g], 0), sc.n_tris - 1);  // the clamps never bind
          const rtmi_triangle& tr = sc.tris[ti];
          double t;
          d3 P;
          bool mine = true;
          if constexpr (EXT) mine = !(WX->member_owner && WX->only_owner >= 0) || WX->member_owner[g] == WX->only_owner;
          if (mine && rtx::tri_hit(rtx::ld3(tr.v[0]), rtx::ld3(tr.v[1]), rtx::ld3(tr.v[2]), sc.planes[ti], o, d, &t, &P)) {
            if (!nearest || best == 0x7fffffff || t < best_t || (t == best_t && g < best)) {
              best = g;
              best_t = t;
              *t_out = t;
              *P_out = P;
            }
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
  *n_tests += 8u * leaf_visits;  // (ray, triangle) sphere tests this lane's ray took part in
  return best == 0x7fffffff ? -1 : best;
}

// Nearest-hit policy: traceRay over the whole top-level list for a secondary
// ray: intersections[0] if want_first, and whether any non-bounding object is hit.
template <bool HAS_TOP>
__device__ inline bool trace_top_secondary_nearest(const SceneDev& sc, const TopTreeRef& tt, d3 o, d3 d, bool want_first, Hit* h,
                                           bool* any_non_bounding, tally_t* n_tests) {
  bool found = false, anb = false;
  constexpr bool nearest = true;  // want_first only: the smallest parameter wins, list order among equals
  // spheres, bounding spheres and lights first, in list order: for a shadow ray the first
  // non-bounding hit settles the question and the triangles need not be searched
  int first_obj = 0x7fffffff;  // list index of the first sphere-like hit (want_first)
#pragma nounroll
  for (int k = 0; k < sc.n_objects; ++k) {
    if (want_first ? (!nearest && found) : anb) break;
    const rtmi_object& ob = sc.objects[k];
    if (ob.kind == RTMI_SPHERE || ob.kind == RTMI_BOUNDING_SPHERE) {
      double t;
      d3 P, n;
      if (want_first) {
        if (rtx::sphere_first_raw(rtx::ld3(ob.center), ob.radius, o, d, &t, &P, &n) && (!nearest || !found || t < h->t)) {
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
        if (want_first && (!nearest || !found || t < h->t)) {
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
      const int g = top_tree_trace_nearest(sc, tt, o, d, !want_first, &t, &P, n_tests);
      if (g >= 0) {
        anb = true;
        const int ti = min(max(sc.top_group[g], 0), sc.n_tris - 1);
        bool take = false;
        if (want_first) {
          const int owner = sc.tri_owner[ti];
          if constexpr (nearest) take = found ? (t < h->t || (t == h->t && owner < first_obj)) : true;
          else take = owner < first_obj;
        }
        if (take) {
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
// shadow-ray extensions (RTMI_FLAG_SHADOW_MAX_DISTANCE, RTMI_FLAG_SHADOWS_ENTER_BOUNDED)
// ---------------------------------------------------------------------------------------
// The scalar part of what the extensions add to a shading launch: the flags and a second bounding-sphere tree,
// over the children of ALL bounding spheres of the scene (built when a frame first asks for
// RTMI_FLAG_SHADOWS_ENTER_BOUNDED; same layout as the tree of the top-level runs).
struct ExtArgs {
  int flags;             // RTMI_FLAG_SHADOW_MAX_DISTANCE | RTMI_FLAG_SHADOWS_ENTER_BOUNDED | RTMI_FLAG_REFLECTIONS_ENTER_BOUNDED
  int pad;
  long long n_bounded;   // members of the bounded tree (0: none, or the flag is off)
  double c0[3];          // centre its FP32 coordinates are relative to
  TopTreeArgs tree;
};

struct ExtRef {
  int flags;
  TopTreeRef bounded;       // tree over the bounded children
  const int* members;       // member ordinal -> triangle index
  const int* member_owner;  // member ordinal -> object index of its bounding sphere
  long long n_bounded;
  double c0[3];
};

__device__ inline ExtRef make_ext_ref(const ExtArgs& ea, const f2* brec, const float* bgst2, const int* bslot_pos,
                                      const int* bmembers, const int* bowner) {
  ExtRef ext;
  ext.flags = ea.flags;
  ext.bounded = make_top_tree(ea.tree, brec, bgst2, bslot_pos);
  ext.members = bmembers;
  ext.member_owner = bowner;
  ext.n_bounded = ea.n_bounded;
  ext.c0[0] = ea.c0[0]; ext.c0[1] = ea.c0[1]; ext.c0[2] = ea.c0[2];
  return ext;
}

// The shadow test RTF:236-243 with the extensions: does anything other than a bounding sphere intersect the
// ray at a parameter below max_t (INFINITY without RTMI_FLAG_SHADOW_MAX_DISTANCE), counting -- with
// RTMI_FLAG_SHADOWS_ENTER_BOUNDED -- the children of every bounding sphere the ray intersects.
// WALK_TOP / WALK_BOUNDED false: the caller knows that the scene has no top-level runs / that the frame does not
// enter bounding spheres, and the walk is left out of the code.
template <bool WALK_TOP = true, bool WALK_BOUNDED = true>
__device__ inline bool shadow_blocked_ext(const SceneDev& sc, const TopTreeRef& tt, const ExtRef& ext, d3 o, d3 d, double max_t,
                                          tally_t* n_tests) {
#pragma nounroll
  for (int k = 0; k < sc.n_objects; ++k) {
    const rtmi_object& ob = sc.objects[k];
    double t;
    d3 P, n;
    if (ob.kind == RTMI_SPHERE) {
      // the smallest root Sphere.intersections appends (REN:265-271): if that one is not below max_t, none is
      if (rtx::sphere_first_raw(rtx::ld3(ob.center), ob.radius, o, d, &t, &P, &n) && t < max_t) return true;
    } else if (ob.kind == RTMI_LIGHT) {
      if (rtx::light_hit(rtx::ld3(sc.lights[ob.first].loc), o, d, &t, &P) && t < max_t) return true;
    }
  }
  double t;
  d3 P;
  if (WALK_TOP && sc.n_top_tris > 0) {
    WalkExt w;
    w.members = sc.top_group; w.member_owner = nullptr; w.n_members = sc.n_top_tris;
    w.c0[0] = sc.c0[0]; w.c0[1] = sc.c0[1]; w.c0[2] = sc.c0[2];
    w.max_t = max_t;
    w.only_owner = -1;
    if (top_tree_trace<true>(sc, tt, o, d, true, &t, &P, n_tests, &w) >= 0) return true;
  }
  if (WALK_BOUNDED && (ext.flags & RTMI_FLAG_SHADOWS_ENTER_BOUNDED) && ext.n_bounded > 0) {
    WalkExt w;
    w.members = ext.members; w.member_owner = ext.member_owner; w.n_members = ext.n_bounded;
    w.c0[0] = ext.c0[0]; w.c0[1] = ext.c0[1]; w.c0[2] = ext.c0[2];
    w.max_t = max_t;
    w.only_owner = -1;
    if (top_tree_trace<true>(sc, ext.bounded, o, d, true, &t, &P, n_tests, &w) >= 0) return true;
  }
  return false;
}

// RTMI_FLAG_REFLECTIONS_ENTER_BOUNDED: the reflection ray's intersection *h is the bounding sphere objects[k]; its
// children are searched the way raytracePixels searches them for a primary ray (RTF:85-93): their first
// intersection in list order (the nearest under that policy) replaces *h, or the ray has hit nothing.
template <bool NEAREST>
__device__ inline bool reflection_enters_bounded(const SceneDev& sc, const ExtRef& ext, int k, d3 o, d3 d, Hit* h, tally_t* n_tests) {
  WalkExt w;
  w.members = ext.members; w.member_owner = ext.member_owner; w.n_members = ext.n_bounded;
  w.c0[0] = ext.c0[0]; w.c0[1] = ext.c0[1]; w.c0[2] = ext.c0[2];
  w.max_t = INFINITY;
  w.only_owner = k;
  double t;
  d3 P;
  int g;
  if constexpr (NEAREST) g = top_tree_trace_nearest<true>(sc, ext.bounded, o, d, false, &t, &P, n_tests, &w);
  else g = top_tree_trace<true>(sc, ext.bounded, o, d, false, &t, &P, n_tests, &w);
  if (g < 0) return false;
  const int ti = min(max(ext.members[g], 0), sc.n_tris - 1);
  const TriPlane pl = sc.planes[ti];
  h->P = P; h->n = rtx::mk(pl.nx, pl.ny, pl.nz); h->t = t; h->has_normal = 1; h->n_unit = 1;
  h->material = sc.tris[ti].material; h->id = ti;
  return true;
}
