// rtmi_obj.cpp -- OBJ / MTL ingest with the reference's semantics (host only, no GPU).
//
// Restates readObjFile (RayTraceUI/ObjLoader.swift:160-201), readMtlFile (:95-158),
// parseNumbers (:61-63) and createTriangleList (RayTraceUI/ViewController.swift:107-120) and
// produces the rtmi_triangle / rtmi_material arrays that rtmi_scene_create consumes.
//
// Where the reference traps (force-unwrapped parse of a malformed number, a missing MTL key,
// an out-of-range vertex index) this loader returns RTMI_ERR_INVALID_ARGUMENT and describes
// the line in rtmi_last_error().  One deliberate addition: the reference creates every
// Triangle with material nil (ViewController.swift:116) and never reads the material
// dictionary, so a loaded mesh renders black there; here each triangle gets the index of the
// material named by its face's `usemtl` (-1 if that name is not in the MTL file).
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <new>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/rtmi.h"

void rtmi_set_last_error(const std::string& msg);  // rtmi.hip

struct rtmi_obj_mesh {
  std::vector<rtmi_triangle> triangles;
  std::vector<rtmi_material> materials;
  std::vector<std::string> material_names;
  int64_t n_vertices = 0, n_texcoords = 0, n_normals = 0;
};

namespace {

// Swift's String.split(separator:) with omittingEmptySubsequences = true
std::vector<std::string> split(const std::string& s, char sep) {
  std::vector<std::string> out;
  size_t i = 0;
  while (i <= s.size()) {
    size_t j = s.find(sep, i);
    if (j == std::string::npos) j = s.size();
    if (j > i) out.emplace_back(s, i, j - i);
    i = j + 1;
  }
  return out;
}

bool starts_with(const std::string& s, const char* prefix) { return s.compare(0, strlen(prefix), prefix) == 0; }

// CharacterSet.whitespacesAndNewlines for the ASCII range
bool is_space(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f'; }

std::string trim(const std::string& s) {
  size_t a = 0, b = s.size();
  while (a < b && is_space(s[a])) ++a;
  while (b > a && is_space(s[b - 1])) --b;
  return s.substr(a, b - a);
}

// Double(String): the whole token must parse
bool parse_double(const std::string& tok, double* out) {
  if (tok.empty() || is_space(tok[0])) return false;
  errno = 0;
  char* end = nullptr;
  double v = strtod(tok.c_str(), &end);
  if (end != tok.c_str() + tok.size()) return false;
  *out = v;
  return true;
}

// Int(String): optional sign, decimal digits only
bool parse_int(const std::string& tok, long long* out) {
  if (tok.empty()) return false;
  size_t i = (tok[0] == '+' || tok[0] == '-') ? 1 : 0;
  if (i == tok.size()) return false;
  for (size_t k = i; k < tok.size(); ++k)
    if (tok[k] < '0' || tok[k] > '9') return false;
  errno = 0;
  long long v = strtoll(tok.c_str(), nullptr, 10);
  if (errno == ERANGE) return false;
  *out = v;
  return true;
}

// parseNumbers (ObjLoader.swift:61-63): split on " ", drop the first token, convert the rest
template <typename T, typename F>
bool parse_numbers(const std::string& line, F conv, std::vector<T>* out, std::string* bad) {
  std::vector<std::string> toks = split(line, ' ');
  out->clear();
  for (size_t i = 1; i < toks.size(); ++i) {
    T v;
    if (!conv(toks[i], &v)) { *bad = toks[i]; return false; }
    out->push_back(v);
  }
  return true;
}

bool read_file(const char* path, std::string* out) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return false;
  std::ostringstream ss;
  ss << f.rdbuf();
  *out = ss.str();
  return true;
}

double clamp01(double v) { return v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v); }  // RGB.init, Utilities.swift:89-92

int32_t fail(const std::string& msg) {
  rtmi_set_last_error(msg);
  return RTMI_ERR_INVALID_ARGUMENT;
}

// readMtlFile, ObjLoader.swift:95-158
int32_t read_mtl(const char* path, rtmi_obj_mesh* mesh) {
  std::string text;
  if (!read_file(path, &text)) return fail(std::string("cannot read ") + path);
  std::vector<std::string> lines = split(text, '\n');
  for (size_t i = 0; i < lines.size(); ++i) {  // `i = j` inside the Swift loop does not skip lines
    std::string x = trim(lines[i]);
    if (starts_with(x, "#") || x.empty()) continue;
    if (!starts_with(x, "newmtl")) continue;
    std::vector<std::string> head = split(x, ' ');
    if (head.size() < 2) return fail("newmtl without a name: '" + lines[i] + "'");
    bool have[6] = {false, false, false, false, false, false};
    rtmi_material m;
    memset(&m, 0, sizeof(m));
    for (size_t j = i + 1; j < lines.size(); ++j) {
      const std::string& l = lines[j];  // the key tests look at the UNtrimmed line
      std::vector<double> nums;
      std::vector<long long> ints;
      std::string bad;
      if (starts_with(l, "Ns ")) {
        if (!parse_numbers<double>(l, parse_double, &nums, &bad) || nums.empty()) return fail("bad Ns line: '" + l + "'");
        m.Ns = nums[0]; have[0] = true; continue;
      }
      if (starts_with(l, "d ")) {
        if (!parse_numbers<double>(l, parse_double, &nums, &bad) || nums.empty()) return fail("bad d line: '" + l + "'");
        m.d = nums[0]; have[1] = true; continue;
      }
      if (starts_with(l, "illum ")) {
        if (!parse_numbers<long long>(l, parse_int, &ints, &bad) || ints.empty()) return fail("bad illum line: '" + l + "'");
        m.illum = (int32_t)ints[0]; have[2] = true; continue;
      }
      auto rgb = [&](double* dst) -> bool {  // RGB(parseNumbers(line))! needs exactly three numbers
        if (!parse_numbers<double>(l, parse_double, &nums, &bad) || nums.size() != 3) return false;
        for (int k = 0; k < 3; ++k) dst[k] = clamp01(nums[k]);
        return true;
      };
      if (starts_with(l, "Kd ")) { if (!rgb(m.Kd)) return fail("bad Kd line: '" + l + "'"); have[3] = true; continue; }
      if (starts_with(l, "Ka ")) { if (!rgb(m.Ka)) return fail("bad Ka line: '" + l + "'"); have[4] = true; continue; }
      if (starts_with(l, "Ks ")) { if (!rgb(m.Ks)) return fail("bad Ks line: '" + l + "'"); have[5] = true; continue; }
      if (trim(l).empty() || starts_with(l, "newmtl")) break;
    }
    for (bool h : have)
      if (!h) return fail("material '" + head[1] + "' lacks one of Ns, d, illum, Kd, Ka, Ks");
    // dictionary assignment: a repeated name replaces the earlier entry
    size_t k = 0;
    for (; k < mesh->material_names.size(); ++k)
      if (mesh->material_names[k] == head[1]) break;
    if (k == mesh->material_names.size()) {
      mesh->material_names.push_back(head[1]);
      mesh->materials.push_back(m);
    } else {
      mesh->materials[k] = m;
    }
  }
  return RTMI_OK;
}

// readObjFile (ObjLoader.swift:160-201) + createTriangleList (ViewController.swift:107-120)
int32_t read_obj(const char* path, rtmi_obj_mesh* mesh) {
  std::string text;
  if (!read_file(path, &text)) return fail(std::string("cannot read ") + path);
  std::vector<std::string> lines = split(text, '\n');
  std::vector<double> verts;
  struct Face { long long a, b, c; std::string material; };
  std::vector<Face> faces;
  std::string material_name;
  for (const std::string& x : lines) {
    if (starts_with(x, "#")) continue;
    std::vector<double> nums;
    std::vector<long long> ints;
    std::string bad;
    if (starts_with(x, "usemtl ")) {
      std::vector<std::string> toks = split(x, ' ');
      material_name = toks.back();  // `.last!`
      continue;
    }
    if (starts_with(x, "v ")) {
      if (!parse_numbers<double>(x, parse_double, &nums, &bad)) return fail("unparsable number '" + bad + "' in '" + x + "'");
      if (nums.size() < 3) return fail("vertex with fewer than 3 numbers: '" + x + "'");
      verts.insert(verts.end(), nums.begin(), nums.begin() + 3);
      continue;
    }
    if (starts_with(x, "vt ")) {
      if (!parse_numbers<double>(x, parse_double, &nums, &bad)) return fail("unparsable number '" + bad + "' in '" + x + "'");
      if (nums.size() < 2) return fail("vt with fewer than 2 numbers: '" + x + "'");
      mesh->n_texcoords += 1;
      continue;
    }
    if (starts_with(x, "vn ")) {
      if (!parse_numbers<double>(x, parse_double, &nums, &bad)) return fail("unparsable number '" + bad + "' in '" + x + "'");
      if (nums.size() < 3) return fail("vn with fewer than 3 numbers: '" + x + "'");
      mesh->n_normals += 1;
      continue;
    }
    if (starts_with(x, "f ")) {
      // plain integers only: "1/2/3" is not an Int in the reference either
      if (!parse_numbers<long long>(x, parse_int, &ints, &bad)) return fail("face token '" + bad + "' is not an integer in '" + x + "'");
      if (ints.size() < 3) return fail("face with fewer than 3 indices: '" + x + "'");
      faces.push_back({ints[0], ints[1], ints[2], material_name});  // a quad keeps its first three
      continue;
    }
  }
  mesh->n_vertices = (int64_t)(verts.size() / 3);
  mesh->triangles.reserve(faces.size());
  for (const Face& f : faces) {
    long long idx[3] = {f.a - 1, f.b - 1, f.c - 1};  // 1-based -> 0-based; no negative-index support
    rtmi_triangle t;
    memset(&t, 0, sizeof(t));
    for (int k = 0; k < 3; ++k) {
      if (idx[k] < 0 || idx[k] >= mesh->n_vertices) return fail("face references vertex " + std::to_string(idx[k] + 1) + " of " + std::to_string(mesh->n_vertices));
      for (int a = 0; a < 3; ++a) t.v[k][a] = verts[(size_t)idx[k] * 3 + a];
    }
    t.material = -1;
    for (size_t k = 0; k < mesh->material_names.size(); ++k)
      if (mesh->material_names[k] == f.material) t.material = (int32_t)k;
    mesh->triangles.push_back(t);
  }
  return RTMI_OK;
}

}  // namespace

extern "C" int32_t rtmi_obj_load(const char* obj_path, const char* mtl_path, rtmi_obj_mesh** out) {
  if (!obj_path || !out) return RTMI_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  rtmi_obj_mesh* mesh = new (std::nothrow) rtmi_obj_mesh();
  if (!mesh) return RTMI_ERR_OUT_OF_MEMORY;
  int32_t rc = RTMI_OK;
  if (mtl_path && *mtl_path) rc = read_mtl(mtl_path, mesh);  // materials first so faces can refer to them
  if (rc == RTMI_OK) rc = read_obj(obj_path, mesh);
  if (rc != RTMI_OK) {
    delete mesh;
    return rc;
  }
  *out = mesh;
  return RTMI_OK;
}

extern "C" void rtmi_obj_free(rtmi_obj_mesh* mesh) { delete mesh; }

extern "C" int64_t rtmi_obj_triangle_count(const rtmi_obj_mesh* mesh) { return mesh ? (int64_t)mesh->triangles.size() : 0; }

extern "C" const rtmi_triangle* rtmi_obj_triangles(const rtmi_obj_mesh* mesh) {
  return mesh && !mesh->triangles.empty() ? mesh->triangles.data() : nullptr;
}

extern "C" int32_t rtmi_obj_material_count(const rtmi_obj_mesh* mesh) { return mesh ? (int32_t)mesh->materials.size() : 0; }

extern "C" const rtmi_material* rtmi_obj_materials(const rtmi_obj_mesh* mesh) {
  return mesh && !mesh->materials.empty() ? mesh->materials.data() : nullptr;
}

extern "C" const char* rtmi_obj_material_name(const rtmi_obj_mesh* mesh, int32_t index) {
  if (!mesh || index < 0 || index >= (int32_t)mesh->material_names.size()) return nullptr;
  return mesh->material_names[(size_t)index].c_str();
}

extern "C" int64_t rtmi_obj_vertex_count(const rtmi_obj_mesh* mesh) { return mesh ? mesh->n_vertices : 0; }

// loadObjFile (ViewController.swift:84-105) in one call: read the OBJ (and MTL), build the triangle
// list and wrap it in Sphere(boundingObjects:) -- `t = [Sphere(boundingObjects: triangles)]`, a
// bounding sphere without a material -- as the scene's only object, with the caller's lights.
extern "C" int32_t rtmi_scene_from_obj(const char* obj_path, const char* mtl_path, const rtmi_light* lights,
                                       int32_t n_lights, int32_t device, rtmi_scene** out) {
  if (!out) return RTMI_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  if (n_lights < 0 || (n_lights > 0 && !lights)) return RTMI_ERR_INVALID_ARGUMENT;
  rtmi_obj_mesh* mesh = nullptr;
  int32_t rc = rtmi_obj_load(obj_path, mtl_path, &mesh);
  if (rc != RTMI_OK) return rc;
  rtmi_object bound{};
  bound.kind = RTMI_BOUNDING_SPHERE;
  bound.material = -1;
  bound.first = 0;
  bound.count = (int64_t)mesh->triangles.size();
  rc = rtmi_bounding_sphere(mesh->triangles.data(), bound.count, bound.center, &bound.radius);
  if (rc == RTMI_OK) {
    rtmi_scene_desc desc{};
    desc.lights = lights;
    desc.n_lights = n_lights;
    desc.materials = mesh->materials.empty() ? nullptr : mesh->materials.data();
    desc.n_materials = (int32_t)mesh->materials.size();
    desc.objects = &bound;
    desc.n_objects = 1;
    desc.triangles = mesh->triangles.empty() ? nullptr : mesh->triangles.data();
    desc.n_triangles = bound.count;
    rc = rtmi_scene_create(&desc, device, out);
  }
  delete mesh;
  return rc;
}
