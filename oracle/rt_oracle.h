/*
 * rt_oracle.h -- CPU FP64 restatement of nealsid/RayTracer's render core.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the correctness oracle and the CPU baseline: tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may call it; the product library
 * (librtmi.so) never does.
 *
 * Parity pin status: the reference ships no tests, golden vectors or fixtures and cannot be
 * compiled here (Swift/AppKit, no Swift toolchain).  The one output of the reference that exists --
 * the rendered view inside its RayTraceExample.png, two spheres and three lights -- is compared pixel
 * by pixel with this restatement's frame of the same scene (tests/test_reference_screenshot.py); that
 * pins the camera model, the sphere test, list order, calculateLighting for sphere hits, clamping and
 * the 8-bit quantisation to what 1/255 can resolve.  Triangles, bounding spheres, supersampling,
 * lights used as objects and everything below 1/255 rest on known-answer vectors derived by hand
 * from the reference source (tests/test_oracle_kat.py) and on a second, separately written
 * restatement (oracle/swift_walkthrough.py): for those, "parity unpinned" by the reference's own
 * artefacts.  The extension switches (RTMI_HIT_NEAREST, RTMI_FLAG_SHADOW_*, ..._ENTER_BOUNDED) have
 * no counterpart in the reference at all; with them off this file states the reference.
 *
 * Scene/camera/parameter structs are the plain C structs of include/rtmi.h (types only).
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include "../include/rtmi.h"

#ifdef __cplusplus
extern "C" {
#endif

/* getPlaneVectors, RayTracingFunctions.swift:46-57. */
void oracle_plane_vectors(const double dir[3], double u[3], double v[3]);

/* Focal-plane corner ul and pixels-per-world scales: RayTracingFunctions.swift:156-170 and
 * WorldCoordinateSequence.swift:58-59. */
void oracle_frame(const rtmi_camera* cam, int32_t width, int32_t height, double ul[3],
                  double u[3], double v[3], double* ppw_h, double* ppw_v);

/* pixelToWorldCoordinate, WorldCoordinateSequence.swift:62-66. */
void oracle_pixel_to_world(const double ul[3], const double u[3], const double v[3],
                           double ppw_h, double ppw_v, double i, double j, double out[3]);

/* Sub-sample coordinates of pixel (x, y) in WorldCoordinateSequence.next order
 * (WorldCoordinateSequence.swift:74-78).  Writes up to `cap` (i, j) pairs, returns count. */
int32_t oracle_pixel_samples(int32_t x, int32_t y, double subdivision, double* ij, int32_t cap);

/* Sphere.intersections, Renderables.swift:248-282.  Returns number of intersections (0..2);
 * t/P/n in reference order (near root first). */
int32_t oracle_sphere_intersections(const double center[3], double radius, const double o[3],
                                    const double d[3], double t[2], double P[2][3],
                                    double n[2][3]);

/* Triangle.intersections, Renderables.swift:136-167.  Returns 0 or 1. */
int32_t oracle_triangle_intersection(const double v[3][3], const double o[3], const double d[3],
                                     double* t, double P[3], double n[3]);

/* PointLight.intersections, Renderables.swift:80-91 (+ eq3, Utilities.swift:155-157). */
int32_t oracle_light_intersection(const double loc[3], const double o[3], const double d[3],
                                  double* t, double P[3]);

/* Sphere.init(boundingObjects:), Renderables.swift:215-246. */
int32_t oracle_bounding_sphere(const rtmi_triangle* tris, int64_t n, double center[3],
                               double* radius);

/* raytraceWorld + raytracePixels, RayTracingFunctions.swift:145-180 and :59-114, with the
 * same output conventions as rtmi_render (include/rtmi.h).  threads <= 1 renders on the
 * calling thread like the reference (ViewController.swift:154); threads > 1 splits pixel
 * columns over OpenMP threads (results are identical; only the counters are reduced). */
int32_t oracle_render(const rtmi_scene_desc* scene, const rtmi_camera* cam,
                      const rtmi_params* params, uint8_t* rgba, int32_t* hit_id,
                      rtmi_counters* counters, int32_t threads);

#ifdef __cplusplus
}
#endif
#endif
