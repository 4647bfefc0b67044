// Host-code robustness under AddressSanitizer/UBSan: the OBJ/MTL loader, the image writers, the
// bounding-sphere and tree-order helpers, fed with malformed and random input.  Links the host
// sources directly (no GPU code); built and run by tests/test_host_fuzz.py.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../../include/rtmi.h"

static std::string g_err;
void rtmi_set_last_error(const std::string& msg) { g_err = msg; }
extern "C" int32_t rtmi_scene_create(const rtmi_scene_desc*, int32_t, rtmi_scene**) { return RTMI_ERR_NO_DEVICE; }  // not under test
extern "C" int32_t rtmi_bounding_sphere(const rtmi_triangle* tris, int64_t n, double center[3], double* radius) {
  (void)tris; (void)n; center[0] = center[1] = center[2] = 0; *radius = 1; return RTMI_OK;
}

static void write_file(const std::string& path, const std::string& body) {
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) { perror("fopen"); exit(2); }
  fwrite(body.data(), 1, body.size(), f);
  fclose(f);
}

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  const std::string dir = argv[1];
  std::mt19937 rng(12345);
  const char* tokens[] = {"v", "vt", "vn", "f", "usemtl", "newmtl", "Ns", "d", "illum", "Kd", "Ka", "Ks", "1", "2", "3", "-1", "0",
                          "99999999999999999999", "1/2/3", "1e400", "nan", "inf", "-0.0", "0x10", "1.5", "abc", "", " ", "\t", "#",
                          "2147483648", "-2147483649", "1.0e-320", "m"};
  const size_t ntok = sizeof(tokens) / sizeof(tokens[0]);
  int loaded = 0, rejected = 0;
  for (int iter = 0; iter < 3000; ++iter) {
    std::string obj, mtl;
    int lines = 1 + (int)(rng() % 12);
    for (int l = 0; l < lines; ++l) {
      int words = (int)(rng() % 7);
      for (int w = 0; w < words; ++w) { obj += tokens[rng() % ntok]; obj += (rng() % 9 == 0) ? "  " : " "; }
      obj += (rng() % 11 == 0) ? "\r\n" : "\n";
    }
    // a valid prefix now and then, so that later lines are reached
    if (rng() % 2) obj = "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n" + obj;
    lines = (int)(rng() % 10);
    for (int l = 0; l < lines; ++l) {
      int words = (int)(rng() % 6);
      for (int w = 0; w < words; ++w) { mtl += tokens[rng() % ntok]; mtl += " "; }
      mtl += "\n";
    }
    if (rng() % 2) mtl = "newmtl m\nNs 8\nd 1\nillum 2\nKd 1 1 1\nKa 1 1 1\nKs 1 1 1\n\n" + mtl;
    write_file(dir + "/f.obj", obj);
    write_file(dir + "/f.mtl", mtl);
    rtmi_obj_mesh* mesh = nullptr;
    int32_t rc = rtmi_obj_load((dir + "/f.obj").c_str(), (rng() % 3) ? (dir + "/f.mtl").c_str() : nullptr, &mesh);
    if (rc == RTMI_OK) {
      ++loaded;
      // every triangle's material index must be valid
      for (int64_t i = 0; i < rtmi_obj_triangle_count(mesh); ++i) {
        int m = rtmi_obj_triangles(mesh)[i].material;
        if (m < -1 || m >= rtmi_obj_material_count(mesh)) { fprintf(stderr, "bad material index %d\n", m); return 1; }
      }
      for (int32_t i = 0; i < rtmi_obj_material_count(mesh); ++i)
        if (!rtmi_obj_material_name(mesh, i)) return 1;
      rtmi_obj_free(mesh);
    } else {
      ++rejected;
      if (mesh) return 1;
    }
  }
  // image writers: odd sizes, including rows longer than one stored deflate block
  for (int w : {1, 3, 17000}) {
    for (int h : {1, 2}) {
      std::vector<uint8_t> px((size_t)w * h * 4);
      for (auto& b : px) b = (uint8_t)rng();
      if (rtmi_write_png((dir + "/o.png").c_str(), px.data(), w, h) != RTMI_OK) return 1;
      if (rtmi_write_ppm((dir + "/o.ppm").c_str(), px.data(), w, h) != RTMI_OK) return 1;
    }
  }
  if (rtmi_write_png((dir + "/no/such/dir/o.png").c_str(), (const uint8_t*)"abcd", 1, 1) == RTMI_OK) return 1;
  if (rtmi_write_png(nullptr, nullptr, 0, 0) == RTMI_OK) return 1;
  printf("host fuzz ok: %d loaded, %d rejected\n", loaded, rejected);
  return loaded > 100 && rejected > 100 ? 0 : 1;
}
