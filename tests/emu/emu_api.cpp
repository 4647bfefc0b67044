// The C ABI of the CPU-emulated build driven the way a host program drives it, under AddressSanitizer with leak
// detection: TEST INFRASTRUCTURE (tests/test_emulated_kernels.py runs it).  "Device" memory is malloc memory
// here, so a device buffer the library forgets to free is a reported leak, and one it frees twice or uses after
// scene destruction is a reported error.
//   * one scene object rendered at frame sizes that grow and shrink, with windows, tiles, all sample counts and a
//     moving camera (work buffers regrown and reused, cached preparation invalidated), every frame compared with
//     the oracle (oracle/rt_oracle.c, compiled into this program without OpenMP: its C code runs under the
//     same sanitizers) byte for byte;
//   * the begin/end protocol and its errors, destruction with a frame in flight;
//   * two host threads with a scene object each (this part is also run in the ThreadSanitizer build);
//   * every device allocation of scene creation and of a first frame made to fail in turn: the call must report
//     out-of-memory (or succeed), leave nothing behind, and the next call must work.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/rtmi.h"
#include "../../oracle/rt_oracle.h"

#define REQUIRE(c) do { if (!(c)) { printf("emu_api FAILED: %s (line %d) last error: %s\n", #c, __LINE__, rtmi_last_error()); return 1; } } while (0)

static unsigned long long g_rng = 0x2545F4914F6CDD1Dull;
static double rnd() {  // uniform in [-1, 1)
  g_rng ^= g_rng >> 12; g_rng ^= g_rng << 25; g_rng ^= g_rng >> 27;
  return (double)((g_rng * 2685821657736338717ull) >> 11) / 4503599627370496.0 - 1.0;
}

struct Scene {
  std::vector<rtmi_light> lights;
  std::vector<rtmi_material> materials;
  std::vector<rtmi_object> objects;
  std::vector<rtmi_triangle> triangles;
  rtmi_scene_desc desc() const {
    rtmi_scene_desc d{};
    d.lights = lights.data(); d.n_lights = (int32_t)lights.size();
    d.materials = materials.data(); d.n_materials = (int32_t)materials.size();
    d.objects = objects.data(); d.n_objects = (int64_t)objects.size();
    d.triangles = triangles.data(); d.n_triangles = (int64_t)triangles.size();
    return d;
  }
};

static void add_triangles(Scene& s, int n, double cx, double cy, double cz, double spread, double size) {
  for (int i = 0; i < n; ++i) {
    rtmi_triangle t{};
    const double c[3] = {cx + spread * rnd(), cy + spread * rnd(), cz + spread * rnd()};
    for (int v = 0; v < 3; ++v)
      for (int k = 0; k < 3; ++k) t.v[v][k] = c[k] + size * rnd();
    t.material = (int32_t)(i % 2);
    s.triangles.push_back(t);
  }
}

static Scene make_scene() {
  Scene s;
  for (int m = 0; m < 2; ++m) {
    rtmi_material mt{};
    mt.Ns = m ? 3.5 : 8.0; mt.d = 1.0; mt.illum = 2;
    for (int k = 0; k < 3; ++k) { mt.Kd[k] = 0.3 + 0.2 * k + 0.1 * m; mt.Ka[k] = 0.6 - 0.1 * k; mt.Ks[k] = 0.9 - 0.3 * m; }
    s.materials.push_back(mt);
  }
  const double lp[3][3] = {{50, 0, 50}, {-50, 0, -50}, {0, 50, 50}};
  for (int l = 0; l < 3; ++l) {
    rtmi_light lt{};
    for (int k = 0; k < 3; ++k) { lt.loc[k] = lp[l][k]; lt.specular[k] = 0.8; lt.diffuse[k] = 0.5; }
    s.lights.push_back(lt);
  }
  // a bounded group, a plain sphere, a top-level run, a second bounded group
  add_triangles(s, 170, -0.7, 0.0, 0.0, 0.5, 0.3);
  add_triangles(s, 45, 0.0, -0.5, 0.4, 0.5, 0.35);
  add_triangles(s, 70, 0.8, 0.3, -0.4, 0.4, 0.3);
  rtmi_object o{};
  o.kind = RTMI_BOUNDING_SPHERE; o.material = -1; o.first = 0; o.count = 170;
  oracle_bounding_sphere(s.triangles.data(), 170, o.center, &o.radius);
  s.objects.push_back(o);
  o = rtmi_object{};
  o.kind = RTMI_SPHERE; o.material = 1; o.center[0] = 0.1; o.center[1] = 0.8; o.center[2] = -0.9; o.radius = 0.45;
  s.objects.push_back(o);
  o = rtmi_object{};
  o.kind = RTMI_TRIANGLES; o.material = -1; o.first = 170; o.count = 45;
  s.objects.push_back(o);
  o = rtmi_object{};
  o.kind = RTMI_BOUNDING_SPHERE; o.material = 0; o.first = 215; o.count = 70;
  oracle_bounding_sphere(s.triangles.data() + 215, 70, o.center, &o.radius);
  s.objects.push_back(o);
  return s;
}

static rtmi_params make_params(int w, int h, double sub) {
  rtmi_params p{};
  p.width = w; p.height = h; p.pixel_subdivision = sub;
  p.ambient[0] = p.ambient[1] = p.ambient[2] = 0.2;
  return p;
}

static rtmi_camera make_camera(double x, double y, double z) {
  rtmi_camera c{};
  c.pos[0] = x; c.pos[1] = y; c.pos[2] = z;
  c.dir[0] = -x; c.dir[1] = -y; c.dir[2] = -z;  // towards the origin, not normalised
  c.focal_length = 8.0;
  return c;
}

// rtmi_render and the oracle into buffers with the same starting contents: everything must agree.
static int same_as_oracle(rtmi_scene* sc, const Scene& s, const rtmi_camera& cam, const rtmi_params& p) {
  const int spp = rtmi_samples_per_pixel(p.pixel_subdivision);
  REQUIRE(spp > 0);
  const size_t npx = (size_t)p.width * p.height;
  std::vector<uint8_t> a(npx * 4, 0x11), b(npx * 4, 0x11);
  std::vector<int32_t> ha(npx * spp, -9), hb(npx * spp, -9);
  rtmi_counters ca{}, cb{};
  const rtmi_scene_desc d = s.desc();
  REQUIRE(rtmi_render(sc, &cam, &p, a.data(), ha.data(), &ca) == RTMI_OK);
  REQUIRE(oracle_render(&d, &cam, &p, b.data(), hb.data(), &cb, 4) == RTMI_OK);
  REQUIRE(ha == hb);
  REQUIRE(a == b);
  REQUIRE(ca.samples == cb.samples && ca.rays_reflection == cb.rays_reflection && ca.rays_shadow == cb.rays_shadow &&
          ca.primary_hits == cb.primary_hits);
  return 0;
}

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  const bool only_threads = argc > 1 && !strcmp(argv[1], "threads");  // the part the race-detection build runs
  const double t_start = now_s();
  const Scene s = make_scene();
  const rtmi_scene_desc d = s.desc();
  rtmi_scene* sc = nullptr;
  REQUIRE(rtmi_scene_create(&d, 0, &sc) == RTMI_OK);

  // 1. one scene object, many frame shapes
  struct Shape { int w, h; double sub; int win[4]; int tile, rank, world; int search; double cam[3]; };
  const Shape shapes[] = {
      {16, 12, 1.0, {0, 0, 0, 0}, 0, 0, 1, 0, {0, 0, 4}},
      {40, 30, 0.5, {0, 0, 0, 0}, 0, 0, 1, 0, {0, 0, 4}},       // grows
      {8, 8, 0.25, {0, 0, 0, 0}, 0, 0, 1, 1, {0, 0, 4}},        // shrinks, other search, same camera
      {33, 21, 0.5, {5, 3, 30, 17}, 0, 0, 1, 0, {1.5, 0.4, 3.5}},  // window, camera moved
      {33, 21, 0.5, {5, 3, 30, 17}, 8, 1, 3, 0, {1.5, 0.4, 3.5}},  // tiles inside the window, camera kept
      {44, 9, 0.4, {0, 0, 0, 0}, 4, 0, 2, 1, {-2.0, 1.0, 3.0}},
      {2, 2, 1.0, {0, 0, 0, 0}, 0, 0, 1, 0, {0, 0, 4}},         // the smallest legal frame: one pixel
      {44, 36, 0.5, {0, 0, 0, 0}, 0, 0, 1, 0, {0.2, 0.1, 0.3}},  // grows again; camera inside the geometry
  };
  for (const Shape& sh : shapes) {
    if (only_threads) break;
    rtmi_params p = make_params(sh.w, sh.h, sh.sub);
    p.x0 = sh.win[0]; p.y0 = sh.win[1]; p.x1 = sh.win[2]; p.y1 = sh.win[3];
    p.tile_size = sh.tile; p.tile_rank = sh.rank; p.tile_world = sh.world;
    p.search = sh.search;
    if (same_as_oracle(sc, s, make_camera(sh.cam[0], sh.cam[1], sh.cam[2]), p)) return 1;
  }

  printf("frame shapes: %.1f s\n", now_s() - t_start);
  // 2. protocol
  if (!only_threads) {
    const rtmi_params p = make_params(20, 14, 0.5);
    const rtmi_camera cam = make_camera(0, 0, 4);
    uint8_t* frame = nullptr;
    REQUIRE(hipMalloc(&frame, (size_t)4 * 20 * 14) == hipSuccess);
    rtmi_counters c{};
    REQUIRE(rtmi_render_end(sc, &c) == RTMI_ERR_INVALID_ARGUMENT);              // nothing in flight
    REQUIRE(rtmi_render_begin(sc, &cam, &p, frame, nullptr) == RTMI_OK);
    REQUIRE(rtmi_render_begin(sc, &cam, &p, frame, nullptr) == RTMI_ERR_INVALID_ARGUMENT);  // one frame per scene
    REQUIRE(rtmi_render_end(sc, &c) == RTMI_OK && c.samples == 19 * 13 * 4);
    REQUIRE(rtmi_render_device(sc, &cam, &p, frame, nullptr, nullptr) == RTMI_OK);  // no hit ids, no counters
    rtmi_params bad = p;
    bad.width = 1;
    REQUIRE(rtmi_render_device(sc, &cam, &bad, frame, nullptr, nullptr) == RTMI_ERR_INVALID_ARGUMENT);
    bad = p; bad.width = (1 << 20) + 1;
    REQUIRE(rtmi_render_device(sc, &cam, &bad, frame, nullptr, nullptr) == RTMI_ERR_UNSUPPORTED);
    bad = p; bad.pixel_subdivision = 0.0;
    REQUIRE(rtmi_render_device(sc, &cam, &bad, frame, nullptr, nullptr) == RTMI_ERR_INVALID_ARGUMENT);
    bad = p; bad.x1 = 25;
    REQUIRE(rtmi_render_device(sc, &cam, &bad, frame, nullptr, nullptr) == RTMI_ERR_INVALID_ARGUMENT);
    bad = p; bad.search = 7;
    REQUIRE(rtmi_render_device(sc, &cam, &bad, frame, nullptr, nullptr) == RTMI_ERR_INVALID_ARGUMENT);
    REQUIRE(rtmi_render_begin(sc, &cam, &p, frame, nullptr) == RTMI_OK);
    rtmi_scene_destroy(sc);  // with a frame in flight: must wait for it, not free under it
    sc = nullptr;
    REQUIRE(hipFree(frame) == hipSuccess);
  }

  printf("protocol: %.1f s\n", now_s() - t_start);
  // 3. every allocation of creation + first frame fails once
  if (!only_threads) {
    const rtmi_params p = make_params(7, 6, 1.0);  // small: the frame is rendered about a hundred times
    const rtmi_camera cam = make_camera(0, 0, 4);
    std::vector<uint8_t> frame((size_t)4 * 7 * 6);
    int failures = 0;
    for (long nth = 0;; ++nth) {
      emu::M().allocations = 0;
      emu::M().fail_allocation = nth;
      rtmi_scene* t = nullptr;
      int32_t rc = rtmi_scene_create(&d, 0, &t);
      if (rc == RTMI_OK) rc = rtmi_render(t, &cam, &p, frame.data(), nullptr, nullptr);
      const bool injected = emu::M().allocations > nth;
      emu::M().fail_allocation = -1;
      if (injected) {
        REQUIRE(rc == RTMI_ERR_OUT_OF_MEMORY);
        ++failures;
        // the object, if it was created, still works once memory is back
        if (t && nth % 3 == 0) REQUIRE(rtmi_render(t, &cam, &p, frame.data(), nullptr, nullptr) == RTMI_OK);
      } else {
        REQUIRE(rc == RTMI_OK);
      }
      if (t) rtmi_scene_destroy(t);
      if (!injected) break;  // the sequence needs fewer than nth + 1 allocations: all of them have been failed
    }
    REQUIRE(failures > 20);
    printf("allocation failures injected: %d\n", failures);
  }
  printf("allocation failures: %.1f s\n", now_s() - t_start);
  // 4. two host threads, each with a scene object of its own, rendering at the same time; a third reads error text
  {
    int failed[2] = {0, 0};
    auto worker = [&](int who) {
      rtmi_scene* t = nullptr;
      if (rtmi_scene_create(&d, 0, &t) != RTMI_OK) { failed[who] = 1; return; }
      for (int k = 0; k < 3 && !failed[who]; ++k) {
        rtmi_params p = make_params(18 + 4 * who, 12 + 2 * k, who ? 0.5 : 1.0);
        p.search = (who + k) & 1;
        failed[who] = same_as_oracle(t, s, make_camera(0.3 * who, 0.2 * k, 4.0), p);
        rtmi_params bad = p;
        bad.width = 0;  // an error on this thread must not disturb the other's error text
        std::vector<uint8_t> tmp(64);
        if (rtmi_render(t, nullptr, &bad, tmp.data(), nullptr, nullptr) != RTMI_ERR_INVALID_ARGUMENT) failed[who] = 1;
      }
      rtmi_scene_destroy(t);
    };
    std::thread a(worker, 0), b(worker, 1);
    a.join();
    b.join();
    REQUIRE(!failed[0] && !failed[1]);
    printf("two host threads: %.1f s\n", now_s() - t_start);
  }
  if (sc) rtmi_scene_destroy(sc);
  printf("emu_api ok\n");
  return 0;
}
