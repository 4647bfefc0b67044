/*
 * rtmi.h -- C ABI of the MI355X (gfx950) render core for nealsid/RayTracer scenes.
 *
 * The reference exposes no FFI; its seam is the Swift function
 *   raytraceWorld(camera:cameraDirection:focalLength:imageWidthPixels:imageHeightPixels:
 *                 ambientLight:lights:objects:materialDictionary:outputBitmap:pixelDone:)
 * (RayTraceUI/RayTracingFunctions.swift:145-180) and, below it, raytracePixels
 * (RayTracingFunctions.swift:59-114).  Every entry point here replaces a piece of that
 * seam; the citation on each declaration names the reference code it stands in for.
 *
 * All structs are plain-old-data with explicit padding; all reals are IEEE binary64,
 * exactly the values the reference's structs hold.  No torch / C++ types cross this
 * boundary.  Functions return RTMI_OK (0) or a negative rtmi_status.
 */
#ifndef RTMI_H
#define RTMI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTMI_ABI_VERSION 4

typedef enum {
  RTMI_OK = 0,
  RTMI_ERR_INVALID_ARGUMENT = -1, /* null pointer, bad size, bad index                       */
  RTMI_ERR_UNSUPPORTED = -2,      /* legal reference input this build does not handle        */
  RTMI_ERR_NO_DEVICE = -3,        /* no usable gfx950 device / HIP runtime failure at init   */
  RTMI_ERR_HIP = -4,              /* a HIP call failed; see rtmi_last_error()                */
  RTMI_ERR_OUT_OF_MEMORY = -5
} rtmi_status;

/* Camera: the three values typed into the reference UI (ViewController.swift:142-148) plus
 * the focal-length slider (ViewController.swift:38).  dir is NOT normalised by the caller
 * and not by the callee either (RayTracingFunctions.swift:156). */
typedef struct {
  double pos[3];
  double dir[3];
  double focal_length;
} rtmi_camera;

/* PointLight stored properties (Renderables.swift:35-37). */
typedef struct {
  double loc[3];
  double specular[3];
  double diffuse[3];
} rtmi_light;

/* Material (ObjLoader.swift:52-59).  d and illum are carried but unused by shading. */
typedef struct {
  double Ns;
  double d;
  int32_t illum;
  int32_t pad_;
  double Kd[3];
  double Ka[3];
  double Ks[3];
} rtmi_material;

/* Triangle: three explicit FP64 vertices, as createTriangleList builds them
 * (ViewController.swift:107-120), plus a material index (-1 = nil, which is what the
 * reference passes; Renderables.swift:116). */
typedef struct {
  double v[3][3];
  int32_t material;
  int32_t pad_;
} rtmi_triangle;

typedef enum {
  RTMI_SPHERE = 0,          /* Sphere(center, radius)            Renderables.swift:207-213 */
  RTMI_TRIANGLES = 1,       /* a run of `count` Triangle objects occupying consecutive
                               positions of the top-level list   Renderables.swift:115-167 */
  RTMI_BOUNDING_SPHERE = 2, /* Sphere(boundingObjects:)          Renderables.swift:215-246 */
  RTMI_LIGHT = 3            /* a PointLight placed in `objects`  Renderables.swift:80-91   */
} rtmi_kind;

/* One entry of the ordered top-level `objects` array handed to raytraceWorld
 * (RayTracingFunctions.swift:152).  Order is significant: the reference takes the first
 * intersection in list order, not the nearest (RayTracingFunctions.swift:41-43, :83). */
typedef struct {
  int32_t kind;     /* rtmi_kind */
  int32_t material; /* SPHERE / BOUNDING_SPHERE: material index or -1 (nil) */
  double center[3]; /* SPHERE / BOUNDING_SPHERE */
  double radius;    /* SPHERE / BOUNDING_SPHERE; radiusSquared is derived as radius*radius */
  int64_t first;    /* TRIANGLES / BOUNDING_SPHERE: first index into triangles[]; LIGHT: light index */
  int64_t count;    /* TRIANGLES / BOUNDING_SPHERE: number of triangles */
} rtmi_object;

typedef struct {
  const rtmi_light* lights;
  int32_t n_lights;
  int32_t n_materials;
  const rtmi_material* materials;
  const rtmi_object* objects;
  const rtmi_triangle* triangles;
  int64_t n_objects;
  int64_t n_triangles;
} rtmi_scene_desc;

typedef enum {
  RTMI_HIT_LIST_ORDER = 0, /* reference-exact: lowest list position that is hit */
  RTMI_HIT_NEAREST = 1     /* extension, NOT the reference's behaviour: at the three places where the reference takes
                              intersections[0] (RayTracingFunctions.swift:83, :93, :216) the intersection with the
                              smallest parameter is taken, the first in list order among equal parameters; shadow rays
                              (:236-243, "is anything hit") are unaffected.  Cone-tree search only
                              (RTMI_SEARCH_LIST_SCAN: RTMI_ERR_UNSUPPORTED).  A frame with more candidate pairs than
                              the search's lists hold is rendered again with longer lists. */
} rtmi_hit_policy;

/* How the triangles of a group are searched for a primary ray.  Both give the same hits. */
typedef enum {
  RTMI_SEARCH_CONE_TREE = 0, /* default: per-frame cone hierarchy over a spatial order of the
                                triangles; all hits of a ray are found, the lowest list position wins */
  RTMI_SEARCH_LIST_SCAN = 1  /* every triangle a ray from the camera can face, in list order, until the
                                ray's first hit: the reference's own loop (RayTracingFunctions.swift:86-93)
                                behind an FP32 filter */
} rtmi_search;

/* Render parameters.  width/height are the globals imageWidth/imageHeight
 * (ViewController.swift:13-14); the reference renders pixels [0,W-1) x [0,H-1)
 * (RayTracingFunctions.swift:170, WorldCoordinateSequence.swift:69-87).
 * pixel_subdivision is WorldCoordinateSequence.pixelSubdivision (1, 0.5, 0.25 ...).
 * The pixel window and the cyclic tile selection restrict WHICH of those pixels are
 * rendered; pixel->world scaling always uses the full-frame constants.
 * Limits: 2 <= width, height <= 2^20 and at most 2^26 8x8 pixel blocks per frame (larger:
 * RTMI_ERR_UNSUPPORTED); at most 64 sub-samples per axis (4096 per pixel); a subdivision whose
 * stride yields different sample counts at different pixels is refused (RTMI_ERR_UNSUPPORTED). */
typedef struct {
  int32_t width;
  int32_t height;
  double pixel_subdivision;
  double ambient[3];
  int32_t hit_policy;
  int32_t x0, y0, x1, y1; /* window, ends exclusive; all zero = (0,0,W-1,H-1) */
  int32_t tile_size;      /* square tile edge in pixels; 0 = no tiling */
  int32_t tile_rank;      /* render tiles whose id % tile_world == tile_rank */
  int32_t tile_world;     /* <= 1 = every tile */
  int32_t search;         /* rtmi_search */
  int32_t flags;          /* RTMI_FLAG_* bits; 0 = none */
} rtmi_params;

/* The caller wants the pixels (and hit ids), not the reference's ray statistics: secondary rays whose outcome
 * cannot change a pixel need then be neither traced nor counted, and rays_reflection / rays_shadow /
 * shadow_rays_elided of rtmi_counters count only the rays that were traced.  Today this makes a difference for
 * scenes whose top-level list holds nothing but bounding spheres without a material -- what loadObjFile
 * builds (ViewController.swift:101-104): there no reflection or shadow ray can hit anything that blocks light
 * (RayTracingFunctions.swift:241) or carries a material (:192, :245).  The bitmap and the hit ids are the
 * same with and without the flag. */
#define RTMI_FLAG_NO_RAY_STATISTICS 1

/* Extensions to the secondary rays, all off by default and NOT the reference's behaviour.  Two to the shadow test
 * (RayTracingFunctions.swift:234-243 traces a shadow ray against the top-level list only and lets every
 * intersection of a non-bounding object block the light, wherever it lies along the ray):
 * RTMI_FLAG_SHADOW_MAX_DISTANCE    an intersection blocks the light only if its parameter is smaller than the
 *                                  distance from the surface point to the light -- sqrt(lenSquared(light.location -
 *                                  surfacePoint)), the length normalize() divides by at :227 -- so an object behind
 *                                  the light (or a light placed in `objects` at its own location) casts no shadow;
 * RTMI_FLAG_SHADOWS_ENTER_BOUNDED  a shadow ray that intersects a bounding sphere is traced against the sphere's
 *                                  children as well, and a child that is hit blocks the light: the reference's own
 *                                  "TODO seems like I'm forgetting to intersect with shapes inside the bounding
 *                                  shape" (:240).
 * and one to the reflection ray (:207-224), which in the reference sees a bounded mesh as its bounding sphere only:
 * RTMI_FLAG_REFLECTIONS_ENTER_BOUNDED  a reflection ray whose intersection (intersections[0], or the nearest under
 *                                  RTMI_HIT_NEAREST) is a bounding sphere goes on to the sphere's children and takes
 *                                  their first intersection, or none -- what raytracePixels does for the primary ray
 *                                  (:85-93).  A bounded mesh then shows in mirrors, itself included.
 * With any of the three every hit of a frame is shaded from a queue after the search, all its secondary rays traced
 * (RTMI_FLAG_NO_RAY_STATISTICS has no effect), and rtmi_render_device_noalloc refuses the frame
 * (RTMI_ERR_UNSUPPORTED): the first frame that asks for one of the ..._ENTER_BOUNDED flags builds a bounding-sphere
 * tree over the children of all bounding spheres and keeps it with the scene; that one call -- rtmi_render_begin
 * included -- waits for the scene's stream, copies the triangles back to the host and builds the tree there before it
 * enqueues anything (a failure on the way leaves the scene as it was, and a later frame tries again). */
#define RTMI_FLAG_SHADOW_MAX_DISTANCE 2
#define RTMI_FLAG_SHADOWS_ENTER_BOUNDED 4
#define RTMI_FLAG_REFLECTIONS_ENTER_BOUNDED 8

/* Work and timing counters of one render call. */
typedef struct {
  int64_t samples;          /* primary rays generated                                    */
  int64_t rays_reflection;  /* RayTracingFunctions.swift:212                             */
  int64_t rays_shadow;      /* RayTracingFunctions.swift:236                             */
  int64_t tri_tests;        /* (ray, triangle) pairs evaluated (culled or skipped pairs are not counted) */
  int64_t sphere_tests;     /* (ray, sphere|light) pairs evaluated                       */
  int64_t tri_exact_tests;  /* pairs that went through the FP64 reference-order test     */
  int64_t primary_hits;     /* samples that were shaded                                  */
  int64_t shadow_rays_elided; /* of rays_shadow: rays from hits on material-less objects,
                                 counted like the reference counts them but not traced, since
                                 nothing they could hit can change the pixel
                                 (RayTracingFunctions.swift:245)                          */
  int64_t node_tests;       /* cone tree only: (ray, inner node) pairs evaluated         */
  int64_t tri_candidates;   /* cone tree only: pairs that passed the triangle cone and went
                               on to the edge test                                       */
  double ms_total;          /* device time of the whole call (HIP events)                */
  double ms_tri_search;     /* device time of the triangle search launches only (cone tree: walk and
                               FP64 decision; list scan: the scan kernel)                */
  int64_t tri_search_launches;
  int64_t tri_tests_primary_kernel; /* pairs evaluated by the common-origin kernel       */
  double ms_primary_kernel; /* accumulated launch time of that kernel alone: the list scan, or the cone tree's
                               walk (k_search_tree) without the FP64 decisions that follow it */
  double ms_shade;          /* device time of the shading launches                       */
  int64_t primary_rays_traced; /* of samples: those whose ray was generated and went on to the sphere tests and the
                                  triangle search; the rest were cleared beforehand by a conservative test that
                                  shows they can hit nothing (whole 8x8 blocks, then single samples)   */
  int64_t hits_decided_fp32;   /* frames of bounding-sphere-only scenes with RTMI_FLAG_NO_RAY_STATISTICS: (ray, triangle)
                                  pairs accepted by conservative FP32 tests alone -- inside the triangle, its front and
                                  the bounding sphere by margins that exceed the FP32 error; every other pair of
                                  such a frame is counted in tri_exact_tests                          */
  int64_t pixels_fixed_fp64;   /* the same frames: pixels whose bytes the FP32 lighting and its error bound could not
                                  settle; every sample of such a pixel was shaded again with the reference's FP64
                                  arithmetic                                                          */
  int64_t tri_edge_tests;      /* cone tree only: (ray, triangle) pairs that went through the FP32 edge test (the
                                  lanes of a wave evaluate it together for a triangle whose cone some ray of theirs passes) */
  int64_t work_buffer_bytes;   /* device memory of the per-frame work buffers (per-sample arrays, candidate lists, tables,
                                  block lists) the scene holds for frames of this kind                 */
} rtmi_counters;

typedef struct rtmi_scene rtmi_scene;

/* ABI version of the loaded library (== RTMI_ABI_VERSION of the header it was built from). */
int32_t rtmi_abi_version(void);

/* Human-readable text for a status code; never NULL. */
const char* rtmi_status_string(int32_t status);

/* Text of the last HIP failure on the calling thread ("" if none). */
const char* rtmi_last_error(void);

/* Number of HIP devices visible to the process (hipGetDeviceCount), or a negative rtmi_status.
 * rtmi_scene_create refuses a device that is not gfx950 (RTMI_ERR_NO_DEVICE). */
int32_t rtmi_device_count(void);

/* Sphere(boundingObjects:) for a triangle list: centre = AABB midpoint, radius = half the
 * largest axis extent (Renderables.swift:215-246 with Triangle.getBounds :169-195).  Pure
 * host arithmetic; writes center[3] and *radius. */
int32_t rtmi_bounding_sphere(const rtmi_triangle* triangles, int64_t n, double center[3],
                             double* radius);

/* The spatial order in which the cone tree stores a triangle list: slot_pos[slot] = list position,
 * -1 for the padding slots of the last leaf; consecutive runs of 8 slots are the tree's leaves, runs
 * of 8^k leaves its inner nodes.  *n_slots receives the slot count (n rounded up to a multiple of 8,
 * at least 8); slot_pos may be NULL to query it.  Host arithmetic only.  No counterpart in the
 * reference, whose only acceleration is Sphere(boundingObjects:) (Renderables.swift:215-246). */
int32_t rtmi_tree_slot_order(const rtmi_triangle* triangles, int64_t n, int32_t* slot_pos,
                             int64_t capacity, int64_t* n_slots);

/* Flatten and upload a scene to `device` (what the reference holds as the `objects`,
 * `lights` arrays and material values at ViewController.swift:155-167).  The descriptor
 * and everything it points to may be freed after the call returns. */
int32_t rtmi_scene_create(const rtmi_scene_desc* desc, int32_t device, rtmi_scene** out);

void rtmi_scene_destroy(rtmi_scene* scene);

/* Every render call below: when the status is not RTMI_OK the contents of the output buffers are undefined (kernels
 * of the frame may have written part of it), except that rtmi_render leaves the caller's host buffers untouched.
 *
 * raytraceWorld (RayTracingFunctions.swift:145-180) into caller-owned HOST buffers.
 *   rgba_out    4*width*height bytes, row-major RGBA8, bytesPerRow = 4*width
 *               (RayTracingFunctions.swift:67,106-111); only rendered pixels are written.
 *   hit_id_out  optional; width*height*spp int32, index (y*width + x)*spp + k with k in
 *               WorldCoordinateSequence sample order (:74-78).  Values: triangle index
 *               into triangles[] for a triangle hit, -2 - object_index for a sphere,
 *               bounding-sphere-as-sphere or light hit, -1 for a miss.  Unrendered
 *               samples are not written.
 *   counters    optional.
 * Synchronous. */
int32_t rtmi_render(rtmi_scene* scene, const rtmi_camera* camera, const rtmi_params* params,
                    uint8_t* rgba_out, int32_t* hit_id_out, rtmi_counters* counters);

/* Same, but rgba_dev / hit_id_dev are DEVICE pointers on the scene's device (e.g. a torch
 * tensor's data_ptr()).  All work is enqueued on the scene's own stream (created non-blocking:
 * it is NOT ordered against the null stream or any stream of the caller) and the call returns
 * after that stream has drained.
 * Stream contract, for this call and for rtmi_render_begin: everything the caller has enqueued
 * that writes or reads rgba_dev / hit_id_dev (a fill, a copy out of the previous frame) must have
 * completed before the call, or be ordered before it with rtmi_scene_wait_event; the buffers must
 * not be touched again until this call / rtmi_render_end has returned. */
int32_t rtmi_render_device(rtmi_scene* scene, const rtmi_camera* camera,
                           const rtmi_params* params, void* rgba_dev, void* hit_id_dev,
                           rtmi_counters* counters);

/* The same frame in two halves, for callers that have other work to enqueue while the GPU
 * renders (e.g. the transfer of the previous frame): rtmi_render_begin enqueues every kernel
 * of the frame on the scene's stream and returns without waiting; rtmi_render_end waits for
 * that stream and fills `counters` (optional).  One frame may be in flight per scene; camera
 * and params may be freed after rtmi_render_begin returns.  No counterpart in the reference
 * (its render call is synchronous, RayTracingFunctions.swift:145-180). */
int32_t rtmi_render_begin(rtmi_scene* scene, const rtmi_camera* camera, const rtmi_params* params,
                          void* rgba_dev, void* hit_id_dev);
int32_t rtmi_render_end(rtmi_scene* scene, rtmi_counters* counters);

/* rtmi_render_device for callers that cannot allocate device memory while they render: code under
 * HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  It has no counterpart in
 * the reference.  The call allocates nothing on the device (no hipMalloc / hipFree, no allocating
 * library routine): every work buffer of the frame -- per-sample outcomes, directions, colours,
 * queues, candidate lists, sample tables, block list, counters -- is a slice of one array that the
 * library holds in static device memory (9 MiB on each GPU it runs on, for the life of the process).
 * Limits, beyond rtmi_render_device's: at most rtmi_noalloc_max_samples() = 65,536 samples, counted
 * as (8x8 pixel blocks the window touches) x 64 x samples per pixel, and
 * (width + height) x sub-samples per axis <= 16,384; a larger frame is refused with
 * RTMI_ERR_UNSUPPORTED and rtmi_last_error() says which limit.  All calls on one GPU share that
 * GPU's array, which keeps the last call's contents; calls of one process that overlap in time on one GPU
 * (from several scenes or threads) are made to take turns by a per-device lock inside the library (the
 * call is synchronous, so the lock costs nothing).  Calls from SEVERAL PROCESSES cannot overlap: each
 * process holds its own copy of the array.  Results are identical to rtmi_render_device's. */
int32_t rtmi_render_device_noalloc(rtmi_scene* scene, const rtmi_camera* camera,
                                   const rtmi_params* params, void* rgba_dev, void* hit_id_dev,
                                   rtmi_counters* counters);
int64_t rtmi_noalloc_max_samples(void);

/* Frames of scenes whose list holds nothing but bounding spheres without a material, rendered with
 * RTMI_FLAG_NO_RAY_STATISTICS, derive their sample directions in FP32 (pixelToWorldCoordinate minus the camera from
 * per-column and per-row tables, WorldCoordinateSequence.swift:62-66, then a normalisation).  *bound receives the bound the
 * library assumes on the distance between such a unit direction and the reference's FP64 one for this camera and frame
 * size -- every FP32 margin of that path is widened by it -- or +infinity if the frame has no usable bound (the general path
 * renders it then).  Host arithmetic only; exported so that the bound can be checked against observed errors
 * (tests/test_fast_path_math.py).  No counterpart in the reference, which computes in FP64 throughout. */
int32_t rtmi_direction_error_bound(const rtmi_camera* camera, int32_t width, int32_t height, double* bound);

/* Orders the scene's stream after work of the caller's: everything enqueued on the scene's stream
 * after this call (the next rtmi_render_begin / rtmi_render_device) starts only when `hip_event`
 * -- a hipEvent_t the caller has recorded on its own stream, on the scene's device -- has
 * completed (hipStreamWaitEvent).  Does not block the host.  No counterpart in the reference,
 * whose render is one synchronous call on one thread (ViewController.swift:154-180). */
int32_t rtmi_scene_wait_event(rtmi_scene* scene, void* hip_event);

/* Number of sub-samples per pixel for a subdivision value: the count of
 * stride(from: x, to: x+1, by: s) squared (WorldCoordinateSequence.swift:74-78). */
int32_t rtmi_samples_per_pixel(double pixel_subdivision);

/* ---- OBJ / MTL ingest (host only) ---------------------------------------------------- */

typedef struct rtmi_obj_mesh rtmi_obj_mesh;

/* readObjFile (ObjLoader.swift:160-201), readMtlFile (:95-158; mtl_path may be NULL) and
 * createTriangleList (ViewController.swift:107-120) with the reference's parsing rules:
 * lines split on "\n", tokens on " ", `f` takes plain 1-based integers and keeps the first
 * three, `usemtl` names the material of the faces that follow, every MTL block needs
 * Ns, d, illum, Kd, Ka, Ks and its colours are clamped to [0,1] like RGB.init.  Input the
 * reference would trap on is reported as RTMI_ERR_INVALID_ARGUMENT.  Unlike the reference,
 * each triangle's material index is set from its face's material name (-1 if unknown). */
int32_t rtmi_obj_load(const char* obj_path, const char* mtl_path, rtmi_obj_mesh** out);
void rtmi_obj_free(rtmi_obj_mesh* mesh);
int64_t rtmi_obj_vertex_count(const rtmi_obj_mesh* mesh);
int64_t rtmi_obj_triangle_count(const rtmi_obj_mesh* mesh);
const rtmi_triangle* rtmi_obj_triangles(const rtmi_obj_mesh* mesh);   /* owned by the mesh */
int32_t rtmi_obj_material_count(const rtmi_obj_mesh* mesh);
const rtmi_material* rtmi_obj_materials(const rtmi_obj_mesh* mesh);   /* owned by the mesh */
const char* rtmi_obj_material_name(const rtmi_obj_mesh* mesh, int32_t index);

/* loadObjFile (ViewController.swift:84-105) in one call: the OBJ's triangles (materials from the MTL,
 * mtl_path may be NULL) wrapped in Sphere(boundingObjects:) -- a bounding sphere without a material, the
 * scene's only object -- plus the caller's lights, uploaded to `device`. */
int32_t rtmi_scene_from_obj(const char* obj_path, const char* mtl_path, const rtmi_light* lights,
                            int32_t n_lights, int32_t device, rtmi_scene** out);

/* ---- image output (host only) ----------------------------------------------------------- */

/* Write the RGBA8 frame (4*width bytes per row, as rtmi_render fills it) as a binary PPM (RGB) or
 * as an 8-bit RGBA PNG.  Stands in for the CGImage the reference builds from outputBitmap to show
 * it (ViewController.swift:197-216). */
int32_t rtmi_write_ppm(const char* path, const uint8_t* rgba, int32_t width, int32_t height);
int32_t rtmi_write_png(const char* path, const uint8_t* rgba, int32_t width, int32_t height);

#ifdef __cplusplus
}
#endif
#endif /* RTMI_H */
