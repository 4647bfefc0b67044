// Renders a dumped scene through the C ABI of the CPU-emulated build (tests/emu/hip/hip_runtime.h):
// TEST INFRASTRUCTURE, used by tests/test_emulated_kernels.py.
//
//   emu_render <dir>
// reads  <dir>/job.bin        int32 n_frames, int32 want_hit_ids, int32 frames_per_scene, int32 entry (0: rtmi_render,
//                             1: rtmi_render_device_noalloc; "device" memory is host memory in this build),
//                             rtmi_params, then n_frames rtmi_camera
//        <dir>/lights.bin, materials.bin, objects.bin, triangles.bin   the arrays of rtmi_scene_desc
// writes <dir>/rgba_<k>.bin, hit_<k>.bin, counters_<k>.bin for frame k, and one "frame k status s"
// line per frame plus the emulator's per-kernel totals on stdout.  A new scene object is created
// every frames_per_scene frames (0: one scene for all), so that both the first-frame and the
// cached-preparation paths of the library run.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <string>
#include <vector>

#include "../../include/rtmi.h"

static std::vector<unsigned char> slurp(const std::string& path) {
  std::vector<unsigned char> v;
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) { fprintf(stderr, "cannot read %s\n", path.c_str()); exit(2); }
  unsigned char buf[65536];
  size_t n;
  while ((n = fread(buf, 1, sizeof buf, f)) > 0) v.insert(v.end(), buf, buf + n);
  fclose(f);
  return v;
}

static void spill(const std::string& path, const void* p, size_t n) {
  FILE* f = fopen(path.c_str(), "wb");
  if (!f || fwrite(p, 1, n, f) != n) { fprintf(stderr, "cannot write %s\n", path.c_str()); exit(2); }
  fclose(f);
}

int main(int argc, char** argv) {
  if (argc != 2) { fprintf(stderr, "usage: emu_render <dir>\n"); return 2; }
  const std::string dir = argv[1];
  const auto job = slurp(dir + "/job.bin");
  const auto lights = slurp(dir + "/lights.bin"), materials = slurp(dir + "/materials.bin");
  const auto objects = slurp(dir + "/objects.bin"), triangles = slurp(dir + "/triangles.bin");
  if (job.size() < 16 + sizeof(rtmi_params)) { fprintf(stderr, "short job.bin\n"); return 2; }
  int32_t head[4];
  memcpy(head, job.data(), 16);
  const int n_frames = head[0], want_hit = head[1], per_scene = head[2], entry = head[3];
  rtmi_params params;
  memcpy(&params, job.data() + 16, sizeof params);
  if (job.size() != 16 + sizeof(rtmi_params) + (size_t)n_frames * sizeof(rtmi_camera)) { fprintf(stderr, "job.bin size mismatch\n"); return 2; }
  std::vector<rtmi_camera> cams(n_frames);
  if (n_frames) memcpy(cams.data(), job.data() + 16 + sizeof params, (size_t)n_frames * sizeof(rtmi_camera));

  rtmi_scene_desc desc{};
  desc.lights = (const rtmi_light*)lights.data();
  desc.n_lights = (int32_t)(lights.size() / sizeof(rtmi_light));
  desc.materials = (const rtmi_material*)materials.data();
  desc.n_materials = (int32_t)(materials.size() / sizeof(rtmi_material));
  desc.objects = (const rtmi_object*)objects.data();
  desc.n_objects = (int64_t)(objects.size() / sizeof(rtmi_object));
  desc.triangles = (const rtmi_triangle*)triangles.data();
  desc.n_triangles = (int64_t)(triangles.size() / sizeof(rtmi_triangle));

  const int spp = rtmi_samples_per_pixel(params.pixel_subdivision);
  const size_t npx = (size_t)params.width * params.height;
  rtmi_scene* scene = nullptr;
  int rc_all = 0;
  for (int k = 0; k < n_frames; ++k) {
    if (!scene || (per_scene > 0 && k % per_scene == 0)) {
      if (scene) rtmi_scene_destroy(scene);
      scene = nullptr;
      int rc = rtmi_scene_create(&desc, 0, &scene);
      if (rc != RTMI_OK) { printf("scene_create status %d %s\n", rc, rtmi_last_error()); return 1; }
    }
    // the caller's buffers: rendered pixels only are written, the rest must keep this pattern
    std::vector<uint8_t> rgba(npx * 4, 0x5A);
    std::vector<int32_t> hit(want_hit ? npx * (size_t)std::max(spp, 1) : 0, -77);
    rtmi_counters cnt{};
    int rc = entry == 1 ? rtmi_render_device_noalloc(scene, &cams[k], &params, rgba.data(), want_hit ? hit.data() : nullptr, &cnt)
                        : rtmi_render(scene, &cams[k], &params, rgba.data(), want_hit ? hit.data() : nullptr, &cnt);
    printf("frame %d status %d %s\n", k, rc, rc ? rtmi_last_error() : "");
    if (rc) rc_all = 1;
    const std::string tag = std::to_string(k);
    spill(dir + "/rgba_" + tag + ".bin", rgba.data(), rgba.size());
    if (want_hit) spill(dir + "/hit_" + tag + ".bin", hit.data(), hit.size() * 4);
    spill(dir + "/counters_" + tag + ".bin", &cnt, sizeof cnt);
  }
  if (scene) rtmi_scene_destroy(scene);
  emu::report(stdout);
  return rc_all;
}
