// kernels_shade.h -- part of rtmi.hip, which includes its parts in a fixed order (each relies on the ones before it).
// CalculateLighting: k_shade, k_shade_candidates; k_resolve.
// Not a stand-alone header.
// ---------------------------------------------------------------------------------------
// k_shade: calculateLighting (RayTracingFunctions.swift:182-271) for every hit sample
// ---------------------------------------------------------------------------------------
#ifndef RTMI_SHADE_LEAN_WAVES
#define RTMI_SHADE_LEAN_WAVES 4
#endif
constexpr int SHADE_LEAN_WAVES = RTMI_SHADE_LEAN_WAVES;  // resident waves per SIMD the lean shading kernel is compiled for
#ifndef RTMI_SHADE_INERT_WAVES
#define RTMI_SHADE_INERT_WAVES 4
#endif
constexpr int SHADE_INERT_WAVES = RTMI_SHADE_INERT_WAVES;
#ifndef RTMI_TRACE_TOP_WAVES
#define RTMI_TRACE_TOP_WAVES 4
#endif
constexpr int TRACE_TOP_WAVES = RTMI_TRACE_TOP_WAVES;  // resident waves per SIMD k_trace_top is compiled for
#ifndef RTMI_TRACE_EXT_WAVES
#define RTMI_TRACE_EXT_WAVES 4
#endif
constexpr int TRACE_EXT_WAVES = RTMI_TRACE_EXT_WAVES;  // the same for its instantiation with the shadow-ray extensions

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
__device__ inline bool spec_pow_is_small_integer(double e) { return e >= 1.0 && e <= 64.0 && e == floor(e); }

__device__ inline double spec_pow_small_integer(double base, double e) {
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

__device__ inline double spec_pow(double base, double e) {
  if (spec_pow_is_small_integer(e)) return spec_pow_small_integer(base, e);
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

// traceRay for a secondary ray under the frame's hit policy (the two functions are separate on purpose: the
// reference's list-order code stays exactly as it was measured).
template <bool HAS_TOP, bool NEAREST>
__device__ __forceinline__ bool trace_secondary(const SceneDev& sc, const TopTreeRef& tt, d3 o, d3 d, bool want_first, Hit* h,
                                                bool* any_non_bounding, tally_t* n_tests) {
  if constexpr (NEAREST) return trace_top_secondary_nearest<HAS_TOP>(sc, tt, o, d, want_first, h, any_non_bounding, n_tests);
  else return trace_top_secondary<HAS_TOP>(sc, tt, o, d, want_first, h, any_non_bounding, n_tests);
}

// The secondary rays of one sample can be traced in one kernel and their outcomes used by another: the tracing
// kernel wants as many resident waves as it can get (a tree walk is a chain of dependent loads), the lighting
// arithmetic wants registers.  TRACE_LIVE traces where the lighting needs the outcome (one kernel does both);
// TRACE_RECORD traces and notes the outcomes in a TraceRec; TRACE_REPLAY takes them from the TraceRec instead of
// tracing.  A shadow ray's outcome is filed under its light and its depth (shadow_key), so a recorder may trace the
// rays in any order, and may trace more than the lighting ends up asking for.  At most TRACE_MAX_LIGHTS lights.
enum { TRACE_LIVE = 0, TRACE_RECORD = 1, TRACE_REPLAY = 2 };
constexpr int TRACE_MAX_LIGHTS = 16;
struct TraceRec {
  int id;         // the reflection ray's intersection: triangle index, -2 - object index, or -1 for none
  double t;       // its parameter
  unsigned bits;  // bit shadow_key(light, depth): that shadow ray hit something other than a bounding sphere
};
// depth 0: the shadow rays of the primary hit; depth 1: those of the hit its reflection ray found
__device__ inline unsigned shadow_key(int light, int depth) { return (unsigned)(light + TRACE_MAX_LIGHTS * depth) & 31u; }

// The reflection ray (RTF:212-216): its first intersection under the frame's hit policy.
// EXT with EXT_ENTER: under RTMI_FLAG_REFLECTIONS_ENTER_BOUNDED an intersection that is a bounding sphere is replaced by
// what the ray finds among the sphere's children (reflection_enters_bounded).
template <int MODE, bool HAS_TOP, bool NEAREST, bool EXT = false, bool EXT_ENTER = true>
__device__ __forceinline__ bool traced_first(const SceneDev& sc, const TopTreeRef& tt, d3 o, d3 d, Hit* h, tally_t* n_tests,
                                             TraceRec* rec, const ExtRef* ext = nullptr) {
  if constexpr (MODE == TRACE_REPLAY) {
    const int id = rec->id;
    if (id == -1) return false;
    // the intersection as the tracing functions build it: point = origin + parameter * direction
    h->id = id;
    h->t = rec->t;
    h->P = rtx::add(o, rtx::smul(rec->t, d));
    if (id >= 0) {
      const TriPlane pl = sc.planes[id];
      h->n = rtx::mk(pl.nx, pl.ny, pl.nz); h->has_normal = 1; h->n_unit = 1; h->material = sc.tris[id].material;
    } else {
      const rtmi_object& ob = sc.objects[-2 - id];
      if (ob.kind == RTMI_LIGHT) {
        h->has_normal = 0; h->n_unit = 1; h->material = -1;
      } else {
        const d3 raw = rtx::sub(h->P, rtx::ld3(ob.center));
        h->n_unit = ob.material >= 0;
        h->n = h->n_unit ? rtx::normalize(raw) : raw; h->has_normal = 1; h->material = ob.material;
      }
    }
    return true;
  } else {
    bool anb;
    bool found = trace_secondary<HAS_TOP, NEAREST>(sc, tt, o, d, true, h, &anb, n_tests);
    if constexpr (EXT && EXT_ENTER) {
      // (a scene whose bounding spheres are all empty has no tree: the walk then visits nothing and the ray has hit nothing)
      if (found && h->id <= -2 && (ext->flags & RTMI_FLAG_REFLECTIONS_ENTER_BOUNDED)) {
        const int k = min(-2 - h->id, sc.n_objects - 1);
        if (sc.objects[k].kind == RTMI_BOUNDING_SPHERE) found = reflection_enters_bounded<NEAREST>(sc, *ext, k, o, d, h, n_tests);
      }
    }
    if constexpr (MODE == TRACE_RECORD) {
      rec->id = found ? h->id : -1;
      rec->t = found ? h->t : 0.0;
    }
    return found;
  }
}

// A shadow ray (RTF:236-243): does it hit anything other than a bounding sphere.
template <int MODE, bool HAS_TOP, bool NEAREST>
__device__ __forceinline__ bool traced_any(const SceneDev& sc, const TopTreeRef& tt, d3 o, d3 d, tally_t* n_tests, TraceRec* rec,
                                           unsigned key) {
  if constexpr (MODE == TRACE_REPLAY) {
    return (rec->bits >> key) & 1u;
  } else {
    Hit dummy;
    bool anb = false;
    trace_secondary<HAS_TOP, NEAREST>(sc, tt, o, d, false, &dummy, &anb, n_tests);
    if constexpr (MODE == TRACE_RECORD) rec->bits |= (anb ? 1u : 0u) << key;
    return anb;
  }
}

// LEAN (with REFLECT only): the instantiation for the common case that (1) the reflected ray finds nothing
// whose lighting has to be evaluated -- no hit, a light, or an object without a material, whose contribution
// is exactly zero (RTF:192, :245) -- and (2) the specular exponent is a small whole number.  It leaves the
// nested light loop and the general pow() (forty-odd polynomial constants) out of the code, which is what
// lets the kernel keep its registers for more resident waves.  Otherwise nothing is counted, *defer is set
// and the caller hands the sample to the full instantiation.
// INERT (with LEAN only): every member of the top-level list is a bounding sphere without a material, which is
// what loadObjFile builds (ViewController.swift:101-104).  Nothing a reflection or shadow ray can do there
// changes a pixel -- a bounding sphere blocks no light (RTF:241) and, lacking a material, reflects none
// (RTF:192, :245) -- so the rays are neither traced nor counted (the caller asked for pixels, not for the
// reference's ray statistics: RTMI_FLAG_NO_RAY_STATISTICS).
// EXT (with neither LEAN nor INERT; tracing or recording, a replay does not care how the outcomes came about): the
// shadow test of the light loop follows *ext
// (RTMI_FLAG_SHADOW_MAX_DISTANCE, RTMI_FLAG_SHADOWS_ENTER_BOUNDED; shadow_blocked_ext) instead of RTF:236-243.
// EXT_ENTER false: the frame asks for neither ..._ENTER_BOUNDED flag (the walks of the bounded tree are left out).
template <bool REFLECT, bool HAS_TOP, bool LEAN = false, bool INERT = false, bool NEAREST = false, int MODE = TRACE_LIVE,
          bool EXT = false, bool EXT_ENTER = true>
__device__ __forceinline__ RGBd calculate_lighting(const SceneDev& sc, const TopTreeRef& tt, const ShadeConst& fc, const Hit& is, d3 camera,
                                   tally_t* n_refl, tally_t* n_shadow, tally_t* n_tests, tally_t* n_elided, bool* defer = nullptr,
                                   TraceRec* rec = nullptr, const ExtRef* ext = nullptr) {
  static_assert(!EXT || (!LEAN && !INERT && MODE != TRACE_REPLAY), "the extensions trace in the general instantiation only");
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
    if constexpr (LEAN) {
      if (is.has_normal && !spec_pow_is_small_integer(mat.Ns)) {
        *defer = true;
        return I;
      }
    }
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
  // (the pixels-only instantiation normalizes it together with the first lights' directions, see below)
  if ((REFLECT || has_material) && !INERT) surface_to_camera = rtx::normalize(rtx::sub(camera, surface_point));

  if constexpr (REFLECT) {  // RTF:207-224
    d3 incident = rtx::neg(surface_to_camera);
    d3 reflection = rtx::sub(incident, rtx::smul(2 * rtx::dp(incident, normal), normal));
    RGBd contribution = {0.0, 0.0, 0.0};
    Hit rh;
    bool anb;
    if constexpr (INERT) {
      // contribution stays zero
    } else if constexpr (LEAN) {
      tally_t nt = 0;
      if (trace_secondary<HAS_TOP, NEAREST>(sc, tt, surface_point, reflection, true, &rh, &anb, &nt)) {
        if (rh.has_normal && rh.material >= 0) {
          *defer = true;
          return I;
        }
        if (rh.has_normal) {  // no material: its light loop adds nothing, its shadow rays are counted
          const int c = count_facing_lights(sc, rh.P, rh.n, rh.n_unit);
          *n_shadow += c;
          *n_elided += c;
        }
      }
      *n_refl += 1;
      *n_tests += nt;
    } else {
      *n_refl += 1;
      if (traced_first<MODE, HAS_TOP, NEAREST, EXT, EXT_ENTER>(sc, tt, surface_point, reflection, &rh, n_tests, rec, ext)) {
        contribution = calculate_lighting<false, HAS_TOP, false, false, NEAREST, MODE, EXT, EXT_ENTER>(
            sc, tt, fc, rh, surface_point, n_refl, n_shadow, n_tests, n_elided, nullptr, rec, ext);
      }
    }
    I.r = I.r + contribution.r;
    I.g = I.g + contribution.g;
    I.b = I.b + contribution.b;
  }

  if (!has_material) {  // the light loop adds nothing (RTF:245); count its shadow rays
    if constexpr (!INERT) {
      int c = count_facing_lights(sc, surface_point, normal, true);
      *n_shadow += c;
      *n_elided += c;
    }
    I.r = rtx::clamp01(I.r); I.g = rtx::clamp01(I.g); I.b = rtx::clamp01(I.b);  // RTF:269
    return I;
  }
  // one light of the loop RTF:226-268, given the unit vector towards it
  auto one_light = [&](int li, d3 point_to_light) {
    // The light's nine values are fetched (scalar loads) where the loop uses them.  Without the barrier the
    // compiler lifts the loads of every light out of the enclosing per-sample loop and then spills them.
    const rtmi_light& light = sc.lights[li];
    double ndl = rtx::dp(point_to_light, normal);
    if (ndl <= 0) return;
    if constexpr (EXT) {
      *n_shadow += 1;
      // the distance to the light: the length normalize() divided by at RTF:227 (UTL:27-30)
      const double max_t = (ext->flags & RTMI_FLAG_SHADOW_MAX_DISTANCE)
                               ? sqrt(rtx::len_squared(rtx::sub(rtx::ld3(light.loc), surface_point)))
                               : INFINITY;
      const bool blocked = shadow_blocked_ext<HAS_TOP, EXT_ENTER>(sc, tt, *ext, surface_point, point_to_light, max_t, n_tests);
      if constexpr (MODE == TRACE_RECORD) rec->bits |= (blocked ? 1u : 0u) << shadow_key(li, REFLECT ? 0 : 1);
      if (blocked) return;
    } else if constexpr (!INERT) {
      *n_shadow += 1;
      if (traced_any<MODE, HAS_TOP, NEAREST>(sc, tt, surface_point, point_to_light, n_tests, rec, shadow_key(li, REFLECT ? 0 : 1)))
        return;  // RTF:241
    }
    if (!has_material) return; // RTF:245
    double lr = light.diffuse[0], lg = light.diffuse[1], lb = light.diffuse[2];
    lr = ndl * lr; lg = ndl * lg; lb = ndl * lb;                    // scale(normalLightVectorDp)
    lr = mat.Kd[0] * lr; lg = mat.Kd[1] * lg; lb = mat.Kd[2] * lb;  // scale(material.diffuse)
    I.r = I.r + lr; I.g = I.g + lg; I.b = I.b + lb;
    d3 light_reflection = rtx::normalize(rtx::sub(rtx::smul(2 * ndl, normal), point_to_light));  // RTF:251
    double rv = rtx::dp(light_reflection, surface_to_camera);
    if (rv <= 0) return;  // RTF:255-257
    double specular_term = LEAN ? spec_pow_small_integer(rv, mat.Ns) : spec_pow(rv, mat.Ns);  // RTF:259
    double sr = light.specular[0], sg = light.specular[1], sb = light.specular[2];
    sr = specular_term * sr; sg = specular_term * sg; sb = specular_term * sb;
    sr = mat.Ks[0] * sr; sg = mat.Ks[1] * sg; sb = mat.Ks[2] * sb;
    sr = rtx::clamp01(sr); sg = rtx::clamp01(sg); sb = rtx::clamp01(sb);  // RTF:263
    I.r = I.r + sr; I.g = I.g + sg; I.b = I.b + sb;
  };
  if constexpr (INERT) {
    // Three lights at a time: their unit vectors are normalized side by side (normalize3) -- the square roots and
    // divisions of a hit are most of its arithmetic and each is a chain of dependent instructions -- then the lights
    // are evaluated one after the other as the reference does.  (Doing the same for the three reflected-light
    // vectors of RTF:251 was measured slower: the reference has none for a light the surface faces away from.)
    if (sc.n_lights > 0) {  // the first three lights, and the vector to the camera with them
      d3 v[4], u[4];
#pragma unroll
      for (int k = 0; k < 3; ++k) v[k] = rtx::sub(rtx::ld3(sc.lights[min(k, sc.n_lights - 1)].loc), surface_point);
      v[3] = rtx::sub(camera, surface_point);
      rtx::normalize_n<4>(v, u);
      surface_to_camera = u[3];
      one_light(0, u[0]);
      if (1 < sc.n_lights) one_light(1, u[1]);
      if (2 < sc.n_lights) one_light(2, u[2]);
    }
#pragma nounroll
    for (int l0 = 3; l0 < sc.n_lights; l0 += 3) {
      d3 v[3], ptl[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) v[k] = rtx::sub(rtx::ld3(sc.lights[min(l0 + k, sc.n_lights - 1)].loc), surface_point);
      rtx::normalize3(v, ptl);
      one_light(l0, ptl[0]);
      if (l0 + 1 < sc.n_lights) one_light(l0 + 1, ptl[1]);
      if (l0 + 2 < sc.n_lights) one_light(l0 + 2, ptl[2]);
    }
  } else {
#pragma nounroll
    for (int li = 0; li < sc.n_lights; ++li)
      one_light(li, rtx::normalize(rtx::sub(rtx::ld3(sc.lights[li].loc), surface_point)));
  }
  I.r = rtx::clamp01(I.r); I.g = rtx::clamp01(I.g); I.b = rtx::clamp01(I.b);  // RTF:269
  return I;
}

// Work tallies of a shading lane, summed per wave at the end of the kernel.
struct ShadeTally {
  tally_t n_refl = 0, n_shadow = 0, n_hit = 0, n_tests = 0, n_elided = 0;
  tally_t n_exact = 0;  // pairs decided with Triangle.intersections by the shading kernel itself (single candidates)
};

__device__ inline void shade_tally_flush(DevCounters* cnt, const ShadeTally& t) {
  wave_count(cnt, C_HITS, t.n_hit);
  wave_count(cnt, C_REFL, t.n_refl);
  wave_count(cnt, C_SHADOW, t.n_shadow);
  wave_count(cnt, C_TESTS_SECONDARY, t.n_tests);
  wave_count(cnt, C_SHADOW_ELIDED, t.n_elided);
  wave_count(cnt, C_EXACT, t.n_exact);
}

// The intersection of sample s's primary ray (direction d) with `id` (triangle index, or -2 - object index) as the
// lighting wants it; t_tri is the triangle's intersection parameter.  False if a sphere-like object is missed after all.
__device__ __forceinline__ bool primary_hit(const SceneDev& sc, d3 cam, d3 d, int id, double t_tri, Hit* h) {
  bool ok = false;
  if (id >= 0) {
    const rtmi_triangle& tr = sc.tris[id];
    const TriPlane pl = sc.planes[id];
    // the search already decided this hit with Triangle.intersections; rebuild its point
    // exactly as Renderables.swift:153 does: origin + intersectionParameter * direction
    ok = true;
    h->t = t_tri;
    h->P = rtx::add(cam, rtx::smul(h->t, d));
    h->n = rtx::mk(pl.nx, pl.ny, pl.nz);
    h->has_normal = 1;
    h->n_unit = 1;
    h->material = tr.material;
    h->id = id;
  } else {
    const rtmi_object& ob = sc.objects[-2 - id];
    if (ob.kind == RTMI_LIGHT) {
      ok = rtx::light_hit(rtx::ld3(sc.lights[ob.first].loc), cam, d, &h->t, &h->P);
      h->has_normal = 0;
      h->n_unit = 1;
      h->material = -1;
    } else {
      ok = rtx::sphere_first(rtx::ld3(ob.center), ob.radius, cam, d, &h->t, &h->P, &h->n);
      h->has_normal = 1;
      h->n_unit = 1;
      h->material = ob.material;
    }
    h->id = id;
  }
  return ok;
}

// Shade sample s whose primary ray (direction d) hit `id` (triangle index, or -2 - object index);
// t_tri is the triangle's intersection parameter.  LEAN: returns true, with nothing counted or stored,
// if the sample needs the full instantiation (see calculate_lighting).
// MODE (see TraceRec): TRACE_RECORD traces the sample's secondary rays into *rec and stores no colour.
template <bool HAS_TOP, bool LEAN = false, bool NEAREST = false, int MODE = TRACE_LIVE, bool EXT = false, bool EXT_ENTER = true>
__device__ __forceinline__ bool shade_sample(const ShadeConst& fc, const SceneDev& sc, const TopTreeRef& tt, int s, d3 d,
                                             int id, double t_tri, double* rgb, ShadeTally* tl, TraceRec* rec = nullptr,
                                             const ExtRef* ext = nullptr) {
  d3 cam = rtx::ld3(fc.cam);
  Hit h;
  const bool ok = primary_hit(sc, cam, d, id, t_tri, &h);
  RGBd c = {0.0, 0.0, 0.0};
  if (ok) {
    bool defer = false;
    static_assert(!LEAN, "the lean lighting needs a wave-uniform material: shade_triangle_lean");
    c = calculate_lighting<true, HAS_TOP, false, false, NEAREST, MODE, EXT, EXT_ENTER>(sc, tt, fc, h, cam, &tl->n_refl, &tl->n_shadow,
                                                                                       &tl->n_tests, &tl->n_elided, &defer, rec, ext);
    tl->n_hit += 1;
  }
  if constexpr (MODE != TRACE_RECORD) {
    rgb[(size_t)s * 3 + 0] = c.r;
    rgb[(size_t)s * 3 + 1] = c.g;
    rgb[(size_t)s * 3 + 2] = c.b;
  }
  return false;
}

// The lean lighting (see calculate_lighting) of a triangle hit in a scene without top-level triangle runs.
// Returns true, with nothing counted or stored, if the sample needs the full instantiation.
template <bool INERT, bool NEAREST>
__device__ __forceinline__ bool shade_triangle_lean(const ShadeConst& fc, const SceneDev& sc, const TopTreeRef& tt, int s, d3 d,
                                                    int ti, double t_tri, double* rgb, ShadeTally* tl) {
  const d3 cam = rtx::ld3(fc.cam);
  const TriPlane pl = sc.planes[ti];
  Hit h;
  h.t = t_tri;
  h.P = rtx::add(cam, rtx::smul(h.t, d));  // Renderables.swift:153: origin + intersectionParameter * direction
  h.n = rtx::mk(pl.nx, pl.ny, pl.nz);
  h.has_normal = 1;
  h.n_unit = 1;
  h.material = sc.tris[ti].material;
  h.id = ti;
  bool defer = false;
  const RGBd c = calculate_lighting<true, false, true, INERT, NEAREST>(sc, tt, fc, h, cam, &tl->n_refl, &tl->n_shadow, &tl->n_tests, &tl->n_elided,
                                                       &defer);
  if (defer) return true;
  tl->n_hit += 1;
  rgb[(size_t)s * 3 + 0] = c.r;
  rgb[(size_t)s * 3 + 1] = c.g;
  rgb[(size_t)s * 3 + 2] = c.b;
  return false;
}

// HAS_TOP: the scene lists triangles at top level, so secondary rays scan them (the code of
// that scan is left out of the instantiation for scenes without such runs).
template <bool HAS_TOP, bool NEAREST = false>
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
    shade_sample<HAS_TOP, false, NEAREST>(fc, sc, tt, s, d, id, id >= 0 ? tval[s] : 0.0, rgb, &tl);
  }
  shade_tally_flush(cnt, tl);
}

// k_shade with the shadow-ray extensions: every hit of the frame comes through the shade queue (see
// k_queue_candidates) and is shaded here by the general instantiation.  The top-level runs' tree is
// walked if the scene has one (its root count is zero otherwise); b* are the bounded tree's arrays.
template <bool NEAREST>
__global__ void __launch_bounds__(256)
k_shade_ext(ShadeConst fc, SceneDev sc, TopTreeArgs ta, const f2* __restrict__ trec, const float* __restrict__ gst2,
            const int* __restrict__ tslot_pos, ExtArgs ea, const f2* __restrict__ brec, const float* __restrict__ bgst2,
            const int* __restrict__ bslot_pos, const int* __restrict__ bmembers, const int* __restrict__ bowner,
            const double* __restrict__ dirs, const double* __restrict__ tval, const int* __restrict__ q_shade,
            const int* __restrict__ nq_ptr, const int* __restrict__ hit_id, double* rgb, DevCounters* cnt) {
  const TopTreeRef tt = make_top_tree(ta, trec, gst2, tslot_pos);
  const ExtRef ext = make_ext_ref(ea, brec, bgst2, bslot_pos, bmembers, bowner);
  ShadeTally tl;
  const int nq = *nq_ptr;
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += gridDim.x * blockDim.x) {
    int s = checked(q_shade[q], sc.n_samples, cnt, 32);
    d3 d = rtx::mk(dirs[(size_t)s * 3 + 0], dirs[(size_t)s * 3 + 1], dirs[(size_t)s * 3 + 2]);
    int id = hit_id[s];
    if (id >= 0) id = checked(id, sc.n_tris, cnt, 33);
    else id = -2 - checked(-2 - id, sc.n_objects, cnt, 34);
    shade_sample<true, false, NEAREST, TRACE_LIVE, true>(fc, sc, tt, s, d, id, id >= 0 ? tval[s] : 0.0, rgb, &tl, nullptr, &ext);
  }
  shade_tally_flush(cnt, tl);
}

// Scenes with top-level triangle runs shade their queued hits in two launches.  k_trace_top traces every secondary
// ray of a hit -- the reflection ray, the shadow rays of the reflected hit, the shadow rays of the hit itself,
// each a walk of the bounding-sphere tree -- and keeps only the outcomes: the reflected intersection (rec_id per
// queue entry, its parameter in the sample's second colour slot) and one bit per shadow ray (rec_bits per queue
// entry).  Without the colour arithmetic it needs half the registers of k_shade<true>, so twice the waves hide
// the latency of the walks.  k_shade_replay then evaluates the lighting from the recorded outcomes without
// tracing anything.  The sphere tests are counted where the rays are traced, hits and rays by k_shade_replay.
// k_trace_top runs the lighting code itself in recording mode.
// EXT: the shadow rays follow the extension flags of `ea` (shadow_blocked_ext; b* are the bounded tree's arrays);
// the replay is the same.  With EXT the kernel also serves scenes without top-level runs (HAS_TOP false) and frames
// that do not enter bounding spheres (EXT_ENTER false): the walks that cannot happen are left out of the code.
template <bool NEAREST, bool EXT = false, bool HAS_TOP = true, bool EXT_ENTER = true>
__global__ void __launch_bounds__(256, EXT ? TRACE_EXT_WAVES : TRACE_TOP_WAVES)
k_trace_top(ShadeConst fc, SceneDev sc, TopTreeArgs ta, const f2* __restrict__ trec, const float* __restrict__ gst2,
            const int* __restrict__ tslot_pos, const double* __restrict__ dirs, const double* __restrict__ tval,
            const int* __restrict__ q_shade, const int* __restrict__ nq_ptr, const int* __restrict__ hit_id, double* rgb,
            DevCounters* cnt, int* rec_id, unsigned* rec_bits, ExtArgs ea, const f2* __restrict__ brec,
            const float* __restrict__ bgst2, const int* __restrict__ bslot_pos, const int* __restrict__ bmembers,
            const int* __restrict__ bowner) {
  const TopTreeRef tt = make_top_tree(ta, trec, gst2, tslot_pos);
  ExtRef ext;
  if constexpr (EXT) ext = make_ext_ref(ea, brec, bgst2, bslot_pos, bmembers, bowner);
  ShadeTally tl;
  const int nq = *nq_ptr;
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += gridDim.x * blockDim.x) {
    int s = checked(q_shade[q], sc.n_samples, cnt, 24);
    d3 d = rtx::mk(dirs[(size_t)s * 3 + 0], dirs[(size_t)s * 3 + 1], dirs[(size_t)s * 3 + 2]);
    int id = hit_id[s];
    if (id >= 0) id = checked(id, sc.n_tris, cnt, 25);
    else id = -2 - checked(-2 - id, sc.n_objects, cnt, 26);
    TraceRec rec;
    rec.id = -1; rec.t = 0.0; rec.bits = 0u;
    shade_sample<HAS_TOP, false, NEAREST, TRACE_RECORD, EXT, EXT_ENTER>(fc, sc, tt, s, d, id, id >= 0 ? tval[s] : 0.0, rgb, &tl,
                                                                        &rec, EXT ? &ext : nullptr);
    rec_id[q] = rec.id;
    rec_bits[q] = rec.bits;
    rgb[(size_t)s * 3 + 1] = rec.t;
  }
  wave_count(cnt, C_TESTS_SECONDARY, tl.n_tests);  // hits and rays are counted by k_shade_replay
}

template <bool NEAREST>
__global__ void __launch_bounds__(256)
k_shade_replay(ShadeConst fc, SceneDev sc, TopTreeArgs ta, const f2* __restrict__ trec, const float* __restrict__ gst2,
               const int* __restrict__ tslot_pos, const double* __restrict__ dirs, const double* __restrict__ tval,
               const int* __restrict__ q_shade, const int* __restrict__ nq_ptr, const int* __restrict__ hit_id, double* rgb,
               DevCounters* cnt, const int* __restrict__ rec_id, const unsigned* __restrict__ rec_bits) {
  const TopTreeRef tt = make_top_tree(ta, trec, gst2, tslot_pos);
  ShadeTally tl;
  const int nq = *nq_ptr;
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += gridDim.x * blockDim.x) {
    int s = checked(q_shade[q], sc.n_samples, cnt, 27);
    d3 d = rtx::mk(dirs[(size_t)s * 3 + 0], dirs[(size_t)s * 3 + 1], dirs[(size_t)s * 3 + 2]);
    int id = hit_id[s];
    if (id >= 0) id = checked(id, sc.n_tris, cnt, 28);
    else id = -2 - checked(-2 - id, sc.n_objects, cnt, 29);
    TraceRec rec;
    rec.id = rec_id[q];
    // a recorded intersection is about to be used as an index
    if (rec.id >= 0) rec.id = checked(rec.id, sc.n_tris, cnt, 30);
    else if (rec.id <= -2) rec.id = -2 - checked(-2 - rec.id, sc.n_objects, cnt, 31);
    rec.t = rgb[(size_t)s * 3 + 1];
    rec.bits = rec_bits[q];
    shade_sample<true, false, NEAREST, TRACE_REPLAY>(fc, sc, tt, s, d, id, id >= 0 ? tval[s] : 0.0, rgb, &tl, &rec);
  }
  shade_tally_flush(cnt, tl);  // n_tests stays zero here: the tests were counted where the rays were traced
}

// The cone tree's last stage for the children of a bounding object: one lane per candidate
// (grid-stride over the NSLOT lists, like k_search_exact).  The candidate whose list position is
// the ray's lowest hit -- k_search_exact has settled that -- is the ray's intersection: its
// triangle index goes to hit_id and the sample is shaded here, without a detour through the
// shade queue.  t is evaluated once more (the same two operations, the same value).  Samples
// without a hit keep the bounding object's marker in hit_id; k_resolve reads that as a miss.
// Without top-level triangle runs the lean instantiation of the lighting runs here; the rare sample whose
// reflected ray finds an object with a material goes to the shade queue (with its t), which k_shade serves
// afterwards with the full instantiation.
template <bool HAS_TOP, bool INERT = false, bool NEAREST = false>
__global__ void __launch_bounds__(256, HAS_TOP ? (NEAREST ? 1 : 2) : (INERT ? SHADE_INERT_WAVES : SHADE_LEAN_WAVES))
k_shade_candidates(ShadeConst fc, SceneDev sc, TopTreeArgs ta, const f2* __restrict__ trec, const float* __restrict__ gst2,
                   const int* __restrict__ tslot_pos, const SearchArgs* __restrict__ ap, double* rgb, int par) {
  DevCounters* cnt = (DevCounters*)as_global(ap->cnt);
  const TopTreeRef tt = make_top_tree(ta, trec, gst2, tslot_pos);
  ShadeTally tl;
  if (!cnt->overflow) {
    const int list = blockIdx.x & (NSLOT - 1), part = blockIdx.x / NSLOT, parts = gridDim.x / NSLOT;
    const int list_cap = ap->cand_cap / NSLOT;
    const int n = min(cnt->cand_n[par][list][0], list_cap);
    const auto* cand = as_global(ap->cand) + (size_t)list * list_cap;
    const auto* dirs = as_global(ap->dirs);
    // as in k_search_exact: the next step's entry and what hangs on it are fetched one step ahead
    const int stride = parts * blockDim.x;
    const auto* group = as_global(ap->group);
    const auto* best_pos = as_global(ap->best_pos);
    auto* hit_id = as_global(ap->hit_id);
    auto* tval = as_global(ap->tval);
    auto* q_shade = as_global(ap->q_shade);
    const int n_group = ap->n_group, n_samples = ap->n_samples, n_tris = ap->n_tris;
    auto entry = [&](int k) {  // candidate k with both indices range-checked
      int2 e = cand[k];
      e.x = checked(e.x, n_samples, cnt, 9);
      e.y = checked(e.y, n_group, cnt, 10);
      return e;
    };
    // every lane of the workgroup makes the same number of steps (the votes below see whole waves)
    int i = part * blockDim.x + threadIdx.x;
    int2 c = i < n ? entry(i) : make_int2(0, 0);
    int ti = i < n ? checked(group[c.y], n_tris, cnt, 11) : 0;
    int bp = i < n ? best_pos[c.x] : -1;
    for (int base = part * blockDim.x; base < n; base += stride) {
      const int inext = i + stride;
      const int2 cn = inext < n ? entry(inext) : make_int2(0, 0);
      const int tin = inext < n ? checked(group[cn.y], n_tris, cnt, 11) : 0;
      const int bpn = inext < n ? best_pos[cn.x] : -1;
      const int s = c.x;
      bool deferred = false;
      double tv = 0.0;
      // bp < 0: the ray's only candidate (bp == ~c.y, SearchArgs::singles), still to be decided; else the candidate
      // must be the ray's lowest hit: another triangle of the ray may be listed earlier, or the pair was no hit
      bool want = i < n && (bp < 0 || bp == c.y);
      d3 d = rtx::mk(0.0, 0.0, 0.0);
      if (want) {
        const auto* dp = dirs + (size_t)s * 3;
        d = rtx::mk(dp[0], dp[1], dp[2]);
        if (bp < 0) {  // Triangle.intersections for this pair, here instead of in k_search_exact
          const rtmi_triangle& tr = sc.tris[ti];
          d3 P;
          want = rtx::tri_hit(rtx::ld3(tr.v[0]), rtx::ld3(tr.v[1]), rtx::ld3(tr.v[2]), sc.planes[ti], rtx::ld3(fc.cam), d, &tv, &P);
          tl.n_exact += 1;
        } else {
          tv = rtx::tri_t(sc.planes[ti], rtx::ld3(fc.cam), d);  // k_search_exact accepted this pair
        }
        if (want) hit_id[s] = ti;
      }
      if constexpr (HAS_TOP) {
        if (want) shade_sample<true, false, NEAREST>(fc, sc, tt, s, d, ti, tv, rgb, &tl);
      } else {
        if (want) deferred = shade_triangle_lean<INERT, NEAREST>(fc, sc, tt, s, d, ti, tv, rgb, &tl);
      }
      if constexpr (!HAS_TOP) {
        // samples that need the full lighting go to the shade queue, one atomic per wave and step; every
        // sample is queued at most once per frame, so the queue (one entry per sample) cannot overflow
        const unsigned long long dm = __ballot(deferred);
        if (dm) {
          const int lane = threadIdx.x & 63;
          const int leader = __ffsll((long long)dm) - 1;
          int qb = 0;
          if (lane == leader) qb = atomicAdd(&cnt->q_shade, __popcll(dm));
          qb = __shfl(qb, leader);
          if (deferred) {
            tval[s] = tv;
            q_shade[qb + __popcll(dm & ((1ull << lane) - 1ull))] = s;
          }
        }
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

// The cone tree's last stage when the frame's hits are shaded from the shade queue (the shadow-ray extensions): the
// candidate that is its ray's intersection gets its triangle index into hit_id and its parameter into tval, as
// k_shade_candidates does it, and the sample is queued.  Nearly every candidate is one, so the appends are staged
// per workgroup in LDS and handed over with one returning atomic per QUEUE_STAGE - 256 entries or more (a returning
// atomic per wave and step on the one counter would take longer than the shading).
constexpr int QUEUE_STAGE = 4096;
__global__ void __launch_bounds__(256)
k_queue_candidates(ShadeConst fc, SceneDev sc, const SearchArgs* __restrict__ ap, int par) {
  __shared__ int stage[QUEUE_STAGE];
  __shared__ int stage_n, stage_base;
  DevCounters* cnt = ap->cnt;
  int* q_shade = ap->q_shade;
  if (threadIdx.x == 0) stage_n = 0;
  __syncthreads();
  auto flush = [&]() {  // reached by every thread of the workgroup
    const int m = stage_n;
    __syncthreads();
    if (threadIdx.x == 0) {
      stage_base = m ? atomicAdd(&cnt->q_shade, m) : 0;
      stage_n = 0;
    }
    __syncthreads();
    // every sample is queued at most once per frame, so the queue (one entry per sample) cannot overflow
    for (int k = threadIdx.x; k < m; k += blockDim.x) q_shade[stage_base + k] = stage[k];
    __syncthreads();
  };
  if (!cnt->overflow) {
    const int list = blockIdx.x & (NSLOT - 1), part = blockIdx.x / NSLOT, parts = gridDim.x / NSLOT;
    const int list_cap = ap->cand_cap / NSLOT;
    const int n = min(cnt->cand_n[par][list][0], list_cap);
    const int2* cand = ap->cand + (size_t)list * list_cap;
    const int stride = parts * blockDim.x;
    for (int base = part * blockDim.x; base < n; base += stride) {  // the same number of steps for the whole workgroup
      const int i = base + threadIdx.x;
      if (i < n) {
        int2 c = cand[i];
        c.x = checked(c.x, ap->n_samples, cnt, 35);
        c.y = checked(c.y, ap->n_group, cnt, 36);
        if (ap->best_pos[c.x] == c.y) {  // else another triangle of this ray is listed earlier, or the pair was no hit
          const int s = c.x;
          const int ti = checked(ap->group[c.y], ap->n_tris, cnt, 37);
          const d3 d = rtx::mk(ap->dirs[(size_t)s * 3 + 0], ap->dirs[(size_t)s * 3 + 1], ap->dirs[(size_t)s * 3 + 2]);
          ap->hit_id[s] = ti;
          ap->tval[s] = rtx::tri_t(sc.planes[ti], rtx::ld3(fc.cam), d);  // k_search_exact accepted this pair
          stage[atomicAdd(&stage_n, 1)] = s;
        }
      }
      __syncthreads();
      const int fill = stage_n;  // read by everyone before anyone appends again: the workgroup decides as one
      __syncthreads();
      if (fill > QUEUE_STAGE - 256) flush();
    }
  }
  flush();
  search_lists_retire(cnt, par);
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
  if (fc.block_live && fc.block_live[b] == 0) {
    // no sample of this block can hit anything (k_block_cull; k_gen left its samples alone): every sample is
    // RGB.black, the average is black and UInt8(255 * 0) = 0
    *reinterpret_cast<unsigned int*>(rgba + (size_t)4 * fc.W * y + (size_t)x * 4) = 255u << 24;
    if (hit_out)
      for (int q = 0; q < fc.spp; ++q) hit_out[((size_t)y * fc.W + x) * fc.spp + q] = -1;
    return;
  }
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
