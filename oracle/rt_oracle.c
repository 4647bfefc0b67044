/*
 * rt_oracle.c -- CPU FP64 restatement of nealsid/RayTracer's render core.
 *
 * TEST INFRASTRUCTURE ONLY (see rt_oracle.h).  Pinned to the reference's one artefact, the
 * rendered view in its screenshot, for what that can resolve (sphere scenes, 1/255); parity
 * unpinned by the reference beyond that: hand-derived known-answer vectors and a second
 * restatement hold the rest (rt_oracle.h says which parts).
 *
 * Every function follows the reference's arithmetic operation by operation: same operand
 * order, true divisions where the reference divides, no fused multiply-add (build with
 * -ffp-contract=off; Swift does not contract a*b+c), sqrt and pow from libm.  The one
 * deliberate substitution: pow(x, 2) (Renderables.swift:210,243,253) is evaluated as x*x.
 * A correctly rounded pow gives the same bits; a libm pow that is off by one ulp for y = 2
 * would differ in the last bit of radiusSquared / delta.
 *
 * Abbreviations for citations (all under /root/reference/RayTraceUI/):
 *   RTF = RayTracingFunctions.swift   REN = Renderables.swift
 *   WCS = WorldCoordinateSequence.swift   UTL = Utilities.swift
 */
#include "rt_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { double x, y, z; } v3;

static inline v3 V(const double a[3]) { v3 r = {a[0], a[1], a[2]}; return r; }
static inline void store(double o[3], v3 a) { o[0] = a.x; o[1] = a.y; o[2] = a.z; }
static inline v3 add(v3 a, v3 b) { v3 r = {a.x + b.x, a.y + b.y, a.z + b.z}; return r; }
static inline v3 sub(v3 a, v3 b) { v3 r = {a.x - b.x, a.y - b.y, a.z - b.z}; return r; }
static inline v3 neg(v3 a) { v3 r = {-a.x, -a.y, -a.z}; return r; }
static inline v3 smul(double s, v3 a) { v3 r = {s * a.x, s * a.y, s * a.z}; return r; }  /* s * v */
static inline v3 muls(v3 a, double s) { v3 r = {a.x * s, a.y * s, a.z * s}; return r; }  /* v * s */
static inline v3 divs(v3 a, double s) { v3 r = {a.x / s, a.y / s, a.z / s}; return r; }  /* v / s */

/* dp, UTL:23-25: (ax*bx + ay*by) + az*bz. */
static inline double dp(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
/* lenSquared, UTL:38-40. */
static inline double len_squared(v3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
/* normalize, UTL:27-30: each component DIVIDED by sqrt(dp(a,a)). */
static inline v3 normalize(v3 a) {
  double length = sqrt(dp(a, a));
  v3 r = {a.x / length, a.y / length, a.z / length};
  return r;
}
/* cross, UTL:32-36. */
static inline v3 cross(v3 a, v3 b) {
  v3 r = {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
  return r;
}
/* clampFunction(0,1), UTL:67-78. */
static inline double clamp01(double v) {
  if (v < 0.0) return 0.0;
  if (v > 1.0) return 1.0;
  return v;
}

/* ---- camera frame ------------------------------------------------------------------- */

void oracle_plane_vectors(const double dir[3], double u_out[3], double v_out[3]) {
  /* RTF:49-54 */
  v3 plane_normal = normalize(neg(V(dir)));
  v3 up0 = {0.0, 1.0, -1.0};
  v3 up = normalize(up0);
  v3 u = normalize(cross(up, plane_normal));
  v3 v = cross(plane_normal, u);
  store(u_out, u);
  store(v_out, v);
}

void oracle_frame(const rtmi_camera* cam, int32_t width, int32_t height, double ul_out[3],
                  double u_out[3], double v_out[3], double* ppw_h, double* ppw_v) {
  /* RTF:156-168 */
  v3 camera = V(cam->pos);
  v3 center = add(camera, smul(cam->focal_length, V(cam->dir)));
  oracle_plane_vectors(cam->dir, u_out, v_out);
  v3 u = V(u_out), v = V(v_out);
  double world_h = 10.0, world_v = 10.0;
  v3 hpc = muls(u, world_h / 2.0);
  v3 vpc = muls(v, world_v / 2.0);
  v3 ul = add(sub(center, hpc), vpc);
  v3 ur = add(add(center, hpc), vpc);
  v3 ll = sub(sub(center, hpc), vpc);
  store(ul_out, ul);
  /* RTF:170 + WCS:58-59: startPixel = 0, endPixel = imageWidth-1 / imageHeight-1 */
  *ppw_h = (double)((width - 1) - 0) / sqrt(len_squared(sub(ur, ul)));
  *ppw_v = (double)((height - 1) - 0) / sqrt(len_squared(sub(ll, ul)));
}

void oracle_pixel_to_world(const double ul[3], const double u[3], const double v[3],
                           double ppw_h, double ppw_v, double i, double j, double out[3]) {
  /* WCS:62-66 */
  v3 uc = divs(muls(V(u), i), ppw_h);
  v3 vc = divs(muls(V(v), j), ppw_v);
  store(out, sub(add(V(ul), uc), vc));
}

/* stride(from: start, to: end, by: s) for Double: element k is start.addingProduct(k, s)
 * (Swift standard library, Strideable._step for floating point), kept while < end. */
static int32_t stride_values(double start, double end, double s, double* out, int32_t cap) {
  int32_t n = 0;
  if (!(s > 0.0)) return 0;
  for (;;) {
    double val = fma((double)n, s, start);
    if (!(val < end)) break;
    if (n < cap) out[n] = val;
    ++n;
    if (n > (1 << 20)) break;
  }
  return n;
}

int32_t oracle_pixel_samples(int32_t x, int32_t y, double subdivision, double* ij, int32_t cap) {
  /* WCS:74-78: x outer, y inner */
  double xs[64], ys[64];
  int32_t nx = stride_values((double)x, (double)(x + 1), subdivision, xs, 64);
  int32_t ny = stride_values((double)y, (double)(y + 1), subdivision, ys, 64);
  if (nx > 64 || ny > 64) return -1;
  int32_t n = 0;
  for (int32_t a = 0; a < nx; ++a)
    for (int32_t b = 0; b < ny; ++b) {
      if (n < cap) { ij[2 * n] = xs[a]; ij[2 * n + 1] = ys[b]; }
      ++n;
    }
  return n;
}

/* ---- object intersection ------------------------------------------------------------- */

int32_t oracle_sphere_intersections(const double center[3], double radius, const double o[3],
                                    const double d[3], double t_out[2], double P_out[2][3],
                                    double n_out[2][3]) {
  /* REN:248-282 */
  v3 origin = V(o), direction = V(d), location = V(center);
  double radius_squared = radius * radius; /* REN:210 / :243, pow(radius, 2) */
  v3 center_to_eye = sub(origin, location);
  double a = -dp(direction, center_to_eye);
  double delta = a * a - (len_squared(center_to_eye) - radius_squared); /* pow(a, 2) */
  if (delta < 0) return 0;
  double sqrtdelta = sqrt(delta);
  double cand[2];
  int32_t nc;
  if (delta == 0) {
    cand[0] = a;
    nc = 1;
  } else {
    double t1 = a + sqrtdelta, t2 = a - sqrtdelta;
    if (t1 < t2) { cand[0] = t1; cand[1] = t2; } else { cand[0] = t2; cand[1] = t1; }
    nc = 2;
  }
  int32_t n = 0;
  for (int32_t k = 0; k < nc; ++k) {
    if (!(cand[k] >= 0.0000001)) continue;
    v3 p = add(origin, smul(cand[k], direction));
    v3 nrm = normalize(sub(p, location));
    t_out[n] = cand[k];
    store(P_out[n], p);
    store(n_out[n], nrm);
    ++n;
  }
  return n;
}

int32_t oracle_triangle_intersection(const double vv[3][3], const double o[3], const double d[3],
                                     double* t_out, double P_out[3], double n_out[3]) {
  /* REN:136-167 */
  v3 v0 = V(vv[0]), v1 = V(vv[1]), v2 = V(vv[2]);
  v3 origin = V(o), direction = V(d);
  v3 v0v1 = sub(v1, v0);
  v3 v0v2 = sub(v2, v0);
  v3 v1v2 = sub(v2, v1);
  v3 v2v0 = neg(v0v2);
  v3 normal = normalize(cross(v0v1, v0v2));
  double plane_constant = dp(normal, v0);
  double nddot = dp(normal, neg(direction));
  if (nddot <= 0) return 0;
  double t = (plane_constant - dp(normal, origin)) / -nddot;
  if (t > 0) {
    v3 point = add(origin, smul(t, direction));
    v3 v0p = sub(point, v0);
    v3 v1p = sub(point, v1);
    v3 v2p = sub(point, v2);
    v3 a = cross(v0v1, v0p);
    v3 b = cross(v1v2, v1p);
    v3 c = cross(v2v0, v2p);
    if (dp(a, normal) >= 0 && dp(b, normal) >= 0 && dp(c, normal) >= 0) {
      *t_out = t;
      store(P_out, point);
      store(n_out, normal);
      return 1;
    }
  }
  return 0;
}

int32_t oracle_light_intersection(const double loc[3], const double o[3], const double d[3],
                                  double* t_out, double P_out[3]) {
  /* REN:80-91; eq3 UTL:155-157 */
  v3 origin = V(o), direction = V(d), location = V(loc);
  v3 num = sub(location, origin);
  v3 t = {num.x / direction.x, num.y / direction.y, num.z / direction.z};
  if (t.x >= 0.0000001 && (!isinf(t.x) && t.x == t.y && t.y == t.z)) {
    *t_out = t.x;
    store(P_out, add(origin, smul(t.x, direction)));
    return 1;
  }
  return 0;
}

int32_t oracle_bounding_sphere(const rtmi_triangle* tris, int64_t n, double center[3],
                               double* radius) {
  /* REN:215-246, with Triangle.getBounds REN:169-195 and setOnCondition UTL:17-21 */
  double min_x = INFINITY, min_y = INFINITY, min_z = INFINITY;
  double max_x = -INFINITY, max_y = -INFINITY, max_z = -INFINITY;
  for (int64_t i = 0; i < n; ++i) {
    double tminx = INFINITY, tminy = INFINITY, tminz = INFINITY;
    double tmaxx = -INFINITY, tmaxy = -INFINITY, tmaxz = -INFINITY;
    for (int k = 0; k < 3; ++k) {
      const double* p = tris[i].v[k];
      if (tminx > p[0]) tminx = p[0];
      if (tminy > p[1]) tminy = p[1];
      if (tminz > p[2]) tminz = p[2];
      if (tmaxx < p[0]) tmaxx = p[0];
      if (tmaxy < p[1]) tmaxy = p[1];
      if (tmaxz < p[2]) tmaxz = p[2];
    }
    if (min_x > tminx) min_x = tminx;
    if (min_y > tminy) min_y = tminy;
    if (min_z > tminz) min_z = tminz;
    if (max_x < tmaxx) max_x = tmaxx;
    if (max_y < tmaxy) max_y = tmaxy;
    if (max_z < tmaxz) max_z = tmaxz;
  }
  double xd = fabs(max_x - min_x), yd = fabs(max_y - min_y), zd = fabs(max_z - min_z);
  double m = xd;
  if (yd > m) m = yd;
  if (zd > m) m = zd;
  *radius = m / 2;
  center[0] = max_x - (xd / 2);
  center[1] = max_y - (yd / 2);
  center[2] = max_z - (zd / 2);
  return RTMI_OK;
}

/* ---- tracing ------------------------------------------------------------------------- */

typedef struct {
  v3 point;
  v3 normal;
  int has_normal;
  double t;
  int32_t material; /* -1 = nil */
  int is_bounding;
  int64_t object;   /* index into scene->objects */
  int64_t triangle; /* index into scene->triangles, or -1 */
} isect;

typedef struct {
  int64_t sphere_like; /* spheres + bounding spheres + lights in the top-level list */
  int64_t top_tris;    /* triangles in the top-level list */
} list_stats;

typedef struct {
  int64_t rays_reflection, rays_shadow, tri_tests, sphere_tests, primary_hits, samples;
} tally;

/* traceRay over the top-level list (RTF:37-44) reduced to what its callers consume:
 * intersections[0] (RTF:83, :216) and whether any non-bounding object was hit (RTF:241).
 * The reference evaluates every object; the counters charge every object accordingly. */
/* Hit policy: RTMI_HIT_LIST_ORDER is the reference (intersections[0], RTF:83, :93, :216).  RTMI_HIT_NEAREST is an
 * extension: at those three places the intersection with the smallest `parameter` is taken instead, the first in
 * list order among equal parameters.  Shadow rays (RTF:236-243: is anything hit at all) are the same in both. */
/* Shadow-ray extensions (both off in the reference; they only act on the any_non_bounding question):
 * max_t -- RTMI_FLAG_SHADOW_MAX_DISTANCE: an intersection only blocks the light if its parameter is below max_t, the
 *          distance from the surface point to the light (INFINITY: the reference, RTF:236-243, where anything ahead of
 *          the surface point blocks, however far behind the light it lies);
 * enter -- RTMI_FLAG_SHADOWS_ENTER_BOUNDED: a bounding sphere the ray intersects has its children tested, and a child
 *          that is hit blocks the light (the reference's own TODO at RTF:240). */
static int trace_top(const rtmi_scene_desc* s, const list_stats* ls, v3 o, v3 d, isect* first,
                     int* any_non_bounding, tally* tl, int nearest, double max_t, int enter) {
  double oo[3], dd[3];
  store(oo, o);
  store(dd, d);
  int found = 0;
  int anb = 0;
  tl->sphere_tests += ls->sphere_like;
  tl->tri_tests += ls->top_tris;
  for (int64_t k = 0; k < s->n_objects; ++k) {
    const rtmi_object* ob = &s->objects[k];
    if (any_non_bounding ? anb : (found && !nearest)) break;
    switch (ob->kind) {
      case RTMI_SPHERE:
      case RTMI_BOUNDING_SPHERE: {
        double t[2], P[2][3], n[2][3];
        int32_t c = oracle_sphere_intersections(ob->center, ob->radius, oo, dd, t, P, n);
        if (c > 0) {
          if (ob->kind == RTMI_SPHERE && t[0] < max_t) anb = 1; /* t[0] is the smallest root that was appended */
          if (ob->kind == RTMI_BOUNDING_SPHERE && enter && any_non_bounding && !anb) {
            tl->tri_tests += ob->count;
            for (int64_t i = ob->first; i < ob->first + ob->count; ++i) {
              double tc, Pc[3], nc[3];
              if (oracle_triangle_intersection(s->triangles[i].v, oo, dd, &tc, Pc, nc) && tc < max_t) {
                anb = 1;
                break;
              }
            }
          }
          if (!found || (nearest && t[0] < first->t)) { /* a sphere's own roots are sorted near first (REN:265-271) */
            found = 1;
            first->point = V(P[0]);
            first->normal = V(n[0]);
            first->has_normal = 1;
            first->t = t[0];
            first->material = ob->material;
            first->is_bounding = ob->kind == RTMI_BOUNDING_SPHERE;
            first->object = k;
            first->triangle = -1;
          }
        }
        break;
      }
      case RTMI_TRIANGLES: {
        for (int64_t i = ob->first; i < ob->first + ob->count; ++i) {
          double t, P[3], n[3];
          if (oracle_triangle_intersection(s->triangles[i].v, oo, dd, &t, P, n)) {
            if (t < max_t) anb = 1;
            if (!found || (nearest && t < first->t)) {
              found = 1;
              first->point = V(P);
              first->normal = V(n);
              first->has_normal = 1;
              first->t = t;
              first->material = s->triangles[i].material;
              first->is_bounding = 0;
              first->object = k;
              first->triangle = i;
            }
            /* later triangles of the run cannot change [0]; a shadow ray goes on until something blocks */
            if (any_non_bounding ? anb : !nearest) break;
          }
        }
        break;
      }
      case RTMI_LIGHT: {
        double t, P[3];
        if (oracle_light_intersection(s->lights[ob->first].loc, oo, dd, &t, P)) {
          if (t < max_t) anb = 1;
          if (!found || (nearest && t < first->t)) {
            found = 1;
            first->point = V(P);
            first->has_normal = 0;
            first->t = t;
            first->material = -1; /* REN:39-43 */
            first->is_bounding = 0;
            first->object = k;
            first->triangle = -1;
          }
        }
        break;
      }
      default:
        break;
    }
  }
  if (any_non_bounding) *any_non_bounding = anb;
  return found;
}

/* traceRay over getBoundedIntersectables() (RTF:86-93). */
static int trace_children(const rtmi_scene_desc* s, const rtmi_object* ob, int64_t obj_index, v3 o,
                          v3 d, isect* first, tally* tl, int nearest) {
  double oo[3], dd[3];
  store(oo, o);
  store(dd, d);
  tl->tri_tests += ob->count;
  int found = 0;
  for (int64_t i = ob->first; i < ob->first + ob->count; ++i) {
    double t, P[3], n[3];
    if (oracle_triangle_intersection(s->triangles[i].v, oo, dd, &t, P, n)) {
      if (found && !(t < first->t)) continue;
      found = 1;
      first->point = V(P);
      first->normal = V(n);
      first->has_normal = 1;
      first->t = t;
      first->material = s->triangles[i].material;
      first->is_bounding = 0;
      first->object = obj_index;
      first->triangle = i;
      if (!nearest) return 1;
    }
  }
  return found;
}

typedef struct { double r, g, b; } rgb;

/* calculateLighting, RTF:182-271. */
static rgb calculate_lighting(const rtmi_scene_desc* s, const list_stats* ls, const isect* is,
                              const double ambient[3], v3 camera, int reflect, tally* tl, int nearest, int flags) {
  v3 surface_point = is->point;
  const rtmi_material* material = is->material >= 0 ? &s->materials[is->material] : NULL;

  rgb I = {0.0, 0.0, 0.0}; /* RTF:192 */
  if (material) {          /* RTF:193-196: I = ambientLight; I.scale(material.ambient) */
    I.r = material->Ka[0] * ambient[0];
    I.g = material->Ka[1] * ambient[1];
    I.b = material->Ka[2] * ambient[2];
  }
  if (!is->has_normal) return I; /* RTF:199-202 */
  v3 normal = is->normal;

  v3 surface_to_camera = normalize(sub(camera, surface_point)); /* RTF:204 */

  if (reflect) { /* RTF:207-224 */
    v3 incident = neg(surface_to_camera);
    v3 reflection = sub(incident, smul(2 * dp(incident, normal), normal));
    rgb contribution = {0.0, 0.0, 0.0};
    isect rh;
    tl->rays_reflection += 1;
    int have = trace_top(s, ls, surface_point, reflection, &rh, NULL, tl, nearest, INFINITY, 0);
    /* extension (RTMI_FLAG_REFLECTIONS_ENTER_BOUNDED): a reflection ray whose intersection is a bounding sphere goes on
     * to the sphere's children, as the primary ray does at RTF:85-93 -- their first intersection, or nothing */
    if (have && (flags & RTMI_FLAG_REFLECTIONS_ENTER_BOUNDED) && rh.is_bounding)
      have = trace_children(s, &s->objects[rh.object], rh.object, surface_point, reflection, &rh, tl, nearest);
    if (have) {
      contribution = calculate_lighting(s, ls, &rh, ambient, surface_point, 0, tl, nearest, flags);
    }
    I.r = I.r + contribution.r;
    I.g = I.g + contribution.g;
    I.b = I.b + contribution.b;
  }

  for (int32_t li = 0; li < s->n_lights; ++li) { /* RTF:226-268 */
    const rtmi_light* light = &s->lights[li];
    v3 to_light = sub(V(light->loc), surface_point);
    v3 point_to_light = normalize(to_light);
    double ndl = dp(point_to_light, normal);
    if (ndl <= 0) continue;
    isect dummy;
    int any_non_bounding = 0;
    tl->rays_shadow += 1;
    /* extension: the distance to the light, the length normalize() divided by (UTL:27-30) */
    const double max_t = (flags & RTMI_FLAG_SHADOW_MAX_DISTANCE) ? sqrt(len_squared(to_light)) : INFINITY;
    trace_top(s, ls, surface_point, point_to_light, &dummy, &any_non_bounding, tl, nearest, max_t,
              (flags & RTMI_FLAG_SHADOWS_ENTER_BOUNDED) != 0);
    if (any_non_bounding) continue; /* RTF:241 */
    if (!material) continue;        /* RTF:245 */
    /* RTF:246-249 */
    rgb lc = {light->diffuse[0], light->diffuse[1], light->diffuse[2]};
    lc.r = ndl * lc.r; lc.g = ndl * lc.g; lc.b = ndl * lc.b;
    lc.r = material->Kd[0] * lc.r; lc.g = material->Kd[1] * lc.g; lc.b = material->Kd[2] * lc.b;
    I.r = I.r + lc.r; I.g = I.g + lc.g; I.b = I.b + lc.b;
    /* RTF:251-257 */
    v3 light_reflection = normalize(sub(smul(2 * ndl, normal), point_to_light));
    double rv = dp(light_reflection, surface_to_camera);
    if (rv <= 0) continue;
    /* RTF:259-266 */
    double specular_term = pow(rv, material->Ns);
    rgb sc = {light->specular[0], light->specular[1], light->specular[2]};
    sc.r = specular_term * sc.r; sc.g = specular_term * sc.g; sc.b = specular_term * sc.b;
    sc.r = material->Ks[0] * sc.r; sc.g = material->Ks[1] * sc.g; sc.b = material->Ks[2] * sc.b;
    sc.r = clamp01(sc.r); sc.g = clamp01(sc.g); sc.b = clamp01(sc.b);
    I.r = I.r + sc.r; I.g = I.g + sc.g; I.b = I.b + sc.b;
  }
  I.r = clamp01(I.r); I.g = clamp01(I.g); I.b = clamp01(I.b); /* RTF:269 */
  return I;
}

static int validate(const rtmi_scene_desc* s, const rtmi_params* p) {
  if (!s || !p) return RTMI_ERR_INVALID_ARGUMENT;
  if (p->width < 2 || p->height < 2) return RTMI_ERR_INVALID_ARGUMENT;
  if (!(p->pixel_subdivision > 0.0) || p->pixel_subdivision > 1.0) return RTMI_ERR_INVALID_ARGUMENT;
  if (p->hit_policy != RTMI_HIT_LIST_ORDER && p->hit_policy != RTMI_HIT_NEAREST) return RTMI_ERR_UNSUPPORTED;
  for (int64_t k = 0; k < s->n_objects; ++k) {
    const rtmi_object* ob = &s->objects[k];
    if (ob->material >= s->n_materials) return RTMI_ERR_INVALID_ARGUMENT;
    switch (ob->kind) {
      case RTMI_SPHERE: break;
      case RTMI_TRIANGLES:
      case RTMI_BOUNDING_SPHERE:
        if (ob->first < 0 || ob->count < 0 || ob->first + ob->count > s->n_triangles)
          return RTMI_ERR_INVALID_ARGUMENT;
        break;
      case RTMI_LIGHT:
        if (ob->first < 0 || ob->first >= s->n_lights) return RTMI_ERR_INVALID_ARGUMENT;
        break;
      default: return RTMI_ERR_INVALID_ARGUMENT;
    }
  }
  for (int64_t i = 0; i < s->n_triangles; ++i)
    if (s->triangles[i].material >= s->n_materials) return RTMI_ERR_INVALID_ARGUMENT;
  return RTMI_OK;
}

int32_t oracle_render(const rtmi_scene_desc* s, const rtmi_camera* cam, const rtmi_params* p,
                      uint8_t* rgba, int32_t* hit_id, rtmi_counters* counters, int32_t threads) {
  int rc = validate(s, p);
  if (rc != RTMI_OK) return rc;
  if (!cam || !rgba) return RTMI_ERR_INVALID_ARGUMENT;

  const int32_t W = p->width, H = p->height;
  int32_t x0 = p->x0, y0 = p->y0, x1 = p->x1, y1 = p->y1;
  if (x0 == 0 && y0 == 0 && x1 == 0 && y1 == 0) { x1 = W - 1; y1 = H - 1; }
  if (x0 < 0 || y0 < 0 || x1 > W - 1 || y1 > H - 1 || x0 > x1 || y0 > y1)
    return RTMI_ERR_INVALID_ARGUMENT;

  double ij0[2 * 64 * 64];
  const int32_t spp = oracle_pixel_samples(0, 0, p->pixel_subdivision, ij0, 64 * 64);
  if (spp <= 0 || spp > 64 * 64) return RTMI_ERR_INVALID_ARGUMENT;

  double ul[3], u[3], v[3], ppw_h, ppw_v;
  oracle_frame(cam, W, H, ul, u, v, &ppw_h, &ppw_v);
  const v3 camera = V(cam->pos);

  list_stats ls = {0, 0};
  for (int64_t k = 0; k < s->n_objects; ++k) {
    if (s->objects[k].kind == RTMI_TRIANGLES) ls.top_tris += s->objects[k].count;
    else ls.sphere_like += 1;
  }

  const int32_t ts = p->tile_size;
  const int32_t tiles_x = ts > 0 ? (W - 1 + ts - 1) / ts : 1;
  const int32_t world = p->tile_world > 1 ? p->tile_world : 1;
  const int32_t bytes_per_row = 4 * W; /* RTF:67 with start 0, end W-1 */

  tally total = {0, 0, 0, 0, 0, 0};
  if (threads < 1) threads = 1;
  const int nearest = p->hit_policy == RTMI_HIT_NEAREST;

#pragma omp parallel num_threads(threads)
  {
    tally tl = {0, 0, 0, 0, 0, 0};
    double* ij = (double*)malloc(sizeof(double) * 2 * 64 * 64);
#pragma omp for schedule(dynamic, 1)
    for (int32_t x = x0; x < x1; ++x) {     /* WCS:68-89: x outer ... */
      for (int32_t y = y0; y < y1; ++y) {   /* ... y inner */
        if (ts > 0 && world > 1) {
          int32_t tile = (y / ts) * tiles_x + (x / ts);
          if (tile % world != p->tile_rank) continue;
        }
        /* the stride's count can differ from pixel (0,0)'s when some k*s lies within a rounding error of 1;
         * the pixel's own samples are traced and averaged (hit ids beyond the frame's count are not recorded) */
        const int32_t npx = oracle_pixel_samples(x, y, p->pixel_subdivision, ij, 64 * 64);
        rgb cumulative = {0.0, 0.0, 0.0}; /* UTL:56 */
        for (int32_t k = 0; k < npx; ++k) { /* RTF:72 */
          double fp[3];
          oracle_pixel_to_world(ul, u, v, ppw_h, ppw_v, ij[2 * k], ij[2 * k + 1], fp);
          v3 c2p = normalize(sub(V(fp), camera)); /* RTF:73 */
          tl.samples += 1;
          rgb value = {0.0, 0.0, 0.0}; /* RGB.black, RTF:79,90 */
          int32_t id = -1;
          isect i1;
          if (trace_top(s, &ls, camera, c2p, &i1, NULL, &tl, nearest, INFINITY, 0)) { /* RTF:76-83 */
            int have = 1;
            if (i1.is_bounding) { /* RTF:85-94 */
              have = trace_children(s, &s->objects[i1.object], i1.object, camera, c2p, &i1, &tl, nearest);
            }
            if (have) {
              tl.primary_hits += 1;
              value = calculate_lighting(s, &ls, &i1, p->ambient, camera, 1, &tl, nearest, p->flags); /* RTF:96 */
              id = i1.triangle >= 0 ? (int32_t)i1.triangle : (int32_t)(-2 - i1.object);
            }
          }
          cumulative.r = cumulative.r + value.r; /* UTL:57-59 */
          cumulative.g = cumulative.g + value.g;
          cumulative.b = cumulative.b + value.b;
          if (hit_id && k < spp) hit_id[((size_t)y * W + x) * spp + k] = id;
        }
        double factor = 1 / (double)npx; /* UTL:61-62 */
        double ar = factor * cumulative.r, ag = factor * cumulative.g, ab = factor * cumulative.b;
        size_t first_byte = (size_t)bytes_per_row * y + (size_t)x * 4; /* RTF:106-107 */
        rgba[first_byte] = (uint8_t)(255 * ar);                          /* RTF:108-111 */
        rgba[first_byte + 1] = (uint8_t)(255 * ag);
        rgba[first_byte + 2] = (uint8_t)(255 * ab);
        rgba[first_byte + 3] = 255;
      }
    }
    free(ij);
#pragma omp critical
    {
      total.samples += tl.samples;
      total.rays_reflection += tl.rays_reflection;
      total.rays_shadow += tl.rays_shadow;
      total.tri_tests += tl.tri_tests;
      total.sphere_tests += tl.sphere_tests;
      total.primary_hits += tl.primary_hits;
    }
  }

  if (counters) {
    memset(counters, 0, sizeof(*counters));
    counters->samples = total.samples;
    counters->rays_reflection = total.rays_reflection;
    counters->rays_shadow = total.rays_shadow;
    counters->tri_tests = total.tri_tests;
    counters->sphere_tests = total.sphere_tests;
    counters->primary_hits = total.primary_hits;
  }
  return RTMI_OK;
}
