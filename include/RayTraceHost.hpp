// RayTraceHost.hpp -- C++ mirror of the reference's Swift types and entry point on top of the
// C ABI of rtmi.h (header only).
//
// The reference's host language is Swift, which this image cannot compile; this header gives
// the same names, argument meaning and (where a C++ program can) error behaviour, so that a
// caller reads like the reference's own call site (RayTraceUI/ViewController.swift:155-179):
//
//     Material m{8, 10, 5, RGB(0.5,0.5,0.5), RGB(0.7,0.7,0.7), RGB(1,1,1)};
//     Sphere s(v3d(25,0,0), 25);  s.material = m;
//     raytraceWorld(camera, cameraDirection, focalLength, imageWidth - 1, imageHeight - 1,
//                   ambientLight, lights, {s, s1}, materialDict, outputBitmap, pixelDone);
//
// Mirrors: v3d (RayTracingFunctions.swift:13-14), RGB (Utilities.swift:80-153), Material
// (ObjLoader.swift:52-59), PointLight / Triangle / Sphere (Renderables.swift:34-304),
// raytraceWorld (RayTracingFunctions.swift:145-180), readObjFile / readMtlFile /
// createTriangleList (ObjLoader.swift:95-201, ViewController.swift:107-120).
#pragma once

#include <array>
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <variant>
#include <vector>

#include "rtmi.h"

namespace raytrace {

// the reference's frame-size globals (ViewController.swift:13-14)
inline int imageWidth = 1000;
inline int imageHeight = 1000;

struct v3d {
  double x = 0, y = 0, z = 0;
  v3d() = default;
  v3d(double x_, double y_, double z_) : x(x_), y(y_), z(z_) {}
};

// RGB: the initialiser clamps to [0,1] (Utilities.swift:89-94); add/scale do not (:111-130)
struct RGB {
  double red, green, blue;
  static double clamper(double v) { return v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v); }
  RGB(double r, double g, double b) : red(clamper(r)), green(clamper(g)), blue(clamper(b)) {}
  void clamp() { red = clamper(red); green = clamper(green); blue = clamper(blue); }
  void add(const RGB& o) { red = red + o.red; green = green + o.green; blue = blue + o.blue; }
  void scale(double f) { red = f * red; green = f * green; blue = f * blue; }
  void scale(const RGB& f) { red = f.red * red; green = f.green * green; blue = f.blue * blue; }
  static RGB zero() { return RGB(0, 0, 0); }
};

struct Material {
  double specularExponent;
  double dissolution;
  int illumination;
  RGB diffuse, ambient, specular;
};

struct PointLight {
  RGB specular;
  v3d location;
  RGB diffuse;
  PointLight(v3d atLocation, RGB specular_, RGB diffuse_) : specular(specular_), location(atLocation), diffuse(diffuse_) {}
};

struct Triangle {
  std::optional<Material> material;
  std::vector<v3d> vertices;
  Triangle(std::vector<v3d> points, std::optional<Material> material_ = std::nullopt)
      : material(std::move(material_)), vertices(std::move(points)) {
    if (vertices.size() != 3) throw std::invalid_argument("Triangle needs 3 points");  // assert, Renderables.swift:123
  }
};

class RtmiError : public std::runtime_error {
 public:
  int32_t status;
  RtmiError(int32_t s, const std::string& where)
      : std::runtime_error(where + ": " + rtmi_status_string(s) + " " + rtmi_last_error()), status(s) {}
};

inline rtmi_triangle toC(const Triangle& t, int32_t material) {
  rtmi_triangle c{};
  for (int k = 0; k < 3; ++k) { c.v[k][0] = t.vertices[k].x; c.v[k][1] = t.vertices[k].y; c.v[k][2] = t.vertices[k].z; }
  c.material = material;
  return c;
}

struct Sphere {
  v3d location;
  double radius;
  bool bounding;
  std::vector<Triangle> boundedShapes;
  std::optional<Material> material;

  Sphere(v3d sphereCenter, double sphereRadius) : location(sphereCenter), radius(sphereRadius), bounding(false) {}

  // Sphere(boundingObjects:), Renderables.swift:215-246
  explicit Sphere(std::vector<Triangle> boundingObjects) : bounding(true), boundedShapes(std::move(boundingObjects)) {
    std::vector<rtmi_triangle> c;
    c.reserve(boundedShapes.size());
    for (const auto& t : boundedShapes) c.push_back(toC(t, -1));
    double center[3];
    int32_t rc = rtmi_bounding_sphere(c.data(), (int64_t)c.size(), center, &radius);
    if (rc != RTMI_OK) throw RtmiError(rc, "rtmi_bounding_sphere");
    location = v3d(center[0], center[1], center[2]);
  }
  bool isBounding() const { return bounding; }
};

// a PointLight is a class in the reference; objects hold a reference to one of `lights`
using WorldObject = std::variant<Sphere, Triangle, std::shared_ptr<PointLight>>;

// What the library offers beyond the reference's behaviour; everything off = the reference.
struct RenderExtensions {
  bool nearestHit = false;            // RTMI_HIT_NEAREST: intersections.min(by: parameter) instead of intersections[0]
  bool shadowMaxDistance = false;     // RTMI_FLAG_SHADOW_MAX_DISTANCE: only what lies before the light casts a shadow
  bool shadowsEnterBounded = false;   // RTMI_FLAG_SHADOWS_ENTER_BOUNDED: the TODO at RayTracingFunctions.swift:240
  bool reflectionsEnterBounded = false;  // RTMI_FLAG_REFLECTIONS_ENTER_BOUNDED: mirrors show bounded meshes
};

inline void raytraceWorld(v3d camera, v3d cameraDirection, double focalLength, int /*imageWidthPixels*/,
                          int /*imageHeightPixels*/,  // both ignored: the reference uses the globals (:170)
                          RGB ambientLight, const std::vector<std::shared_ptr<PointLight>>& lights,
                          const std::vector<WorldObject>& objects, const std::map<std::string, Material>& /*materialDictionary*/,
                          std::vector<uint8_t>& outputBitmap, const std::function<void()>& pixelDone = nullptr,
                          double pixelSubdivision = 1.0 /* :170 hard-codes 1 */, int device = 0,
                          RenderExtensions extensions = {}) {
  if (outputBitmap.size() != (size_t)4 * imageWidth * imageHeight)
    throw std::out_of_range("outputBitmap must hold 4 * imageWidth * imageHeight bytes");  // Swift traps on the index
  std::vector<rtmi_light> cl;
  for (const auto& l : lights)
    cl.push_back({{l->location.x, l->location.y, l->location.z}, {l->specular.red, l->specular.green, l->specular.blue},
                  {l->diffuse.red, l->diffuse.green, l->diffuse.blue}});
  std::vector<rtmi_material> cm;
  std::vector<rtmi_triangle> ct;
  std::vector<rtmi_object> co;
  std::map<std::array<double, 12>, int32_t> seen;  // identical materials share one entry
  auto materialIndex = [&](const std::optional<Material>& m) -> int32_t {
    if (!m) return -1;
    const std::array<double, 12> key = {m->specularExponent, m->dissolution, (double)m->illumination,
                                        m->diffuse.red, m->diffuse.green, m->diffuse.blue,
                                        m->ambient.red, m->ambient.green, m->ambient.blue,
                                        m->specular.red, m->specular.green, m->specular.blue};
    auto it = seen.find(key);
    if (it != seen.end()) return it->second;
    seen[key] = (int32_t)cm.size();
    rtmi_material c{};
    c.Ns = m->specularExponent; c.d = m->dissolution; c.illum = m->illumination;
    c.Kd[0] = m->diffuse.red; c.Kd[1] = m->diffuse.green; c.Kd[2] = m->diffuse.blue;
    c.Ka[0] = m->ambient.red; c.Ka[1] = m->ambient.green; c.Ka[2] = m->ambient.blue;
    c.Ks[0] = m->specular.red; c.Ks[1] = m->specular.green; c.Ks[2] = m->specular.blue;
    cm.push_back(c);
    return (int32_t)cm.size() - 1;
  };
  for (const auto& o : objects) {  // order preserved: the first hit in list order wins
    rtmi_object c{};
    if (const Sphere* s = std::get_if<Sphere>(&o)) {
      c.kind = s->bounding ? RTMI_BOUNDING_SPHERE : RTMI_SPHERE;
      c.material = materialIndex(s->material);
      c.center[0] = s->location.x; c.center[1] = s->location.y; c.center[2] = s->location.z;
      c.radius = s->radius;
      c.first = (int64_t)ct.size();
      for (const auto& t : s->boundedShapes) ct.push_back(toC(t, materialIndex(t.material)));
      c.count = (int64_t)ct.size() - c.first;
    } else if (const Triangle* t = std::get_if<Triangle>(&o)) {
      // consecutive triangles of the list form ONE run: the list order among them is the order of triangles[],
      // and the kernels' per-ray loops over the top-level list then see one entry instead of thousands
      if (!co.empty() && co.back().kind == RTMI_TRIANGLES && co.back().first + co.back().count == (int64_t)ct.size()) {
        ct.push_back(toC(*t, materialIndex(t->material)));
        co.back().count += 1;
        continue;
      }
      c.kind = RTMI_TRIANGLES;
      c.material = -1;
      c.first = (int64_t)ct.size();
      c.count = 1;
      ct.push_back(toC(*t, materialIndex(t->material)));
    } else {
      const auto& l = std::get<std::shared_ptr<PointLight>>(o);
      c.kind = RTMI_LIGHT;
      c.material = -1;
      c.first = -1;
      for (size_t i = 0; i < lights.size(); ++i)
        if (lights[i] == l) c.first = (int64_t)i;
      if (c.first < 0) throw std::invalid_argument("a PointLight in `objects` must be one of `lights`");
    }
    co.push_back(c);
  }
  rtmi_scene_desc desc{};
  desc.lights = cl.data(); desc.n_lights = (int32_t)cl.size();
  desc.materials = cm.data(); desc.n_materials = (int32_t)cm.size();
  desc.objects = co.data(); desc.n_objects = (int64_t)co.size();
  desc.triangles = ct.data(); desc.n_triangles = (int64_t)ct.size();
  rtmi_scene* scene = nullptr;
  int32_t rc = rtmi_scene_create(&desc, device, &scene);
  if (rc != RTMI_OK) throw RtmiError(rc, "rtmi_scene_create");
  rtmi_camera cam{{camera.x, camera.y, camera.z}, {cameraDirection.x, cameraDirection.y, cameraDirection.z}, focalLength};
  rtmi_params p{};
  p.width = imageWidth; p.height = imageHeight;
  p.pixel_subdivision = pixelSubdivision;
  p.ambient[0] = ambientLight.red; p.ambient[1] = ambientLight.green; p.ambient[2] = ambientLight.blue;
  p.hit_policy = extensions.nearestHit ? RTMI_HIT_NEAREST : RTMI_HIT_LIST_ORDER;
  // (the reference keeps no ray counts, so none are asked for: same pixels, and the library may skip rays that cannot
  // change one)
  p.flags = RTMI_FLAG_NO_RAY_STATISTICS | (extensions.shadowMaxDistance ? RTMI_FLAG_SHADOW_MAX_DISTANCE : 0) |
            (extensions.shadowsEnterBounded ? RTMI_FLAG_SHADOWS_ENTER_BOUNDED : 0) |
            (extensions.reflectionsEnterBounded ? RTMI_FLAG_REFLECTIONS_ENTER_BOUNDED : 0);
  rtmi_counters counters{};
  rc = rtmi_render(scene, &cam, &p, outputBitmap.data(), nullptr, &counters);
  rtmi_scene_destroy(scene);
  if (rc != RTMI_OK) throw RtmiError(rc, "rtmi_render");
  if (pixelDone) {
    int64_t spp = rtmi_samples_per_pixel(pixelSubdivision);
    for (int64_t i = 0; i < counters.samples / spp; ++i) pixelDone();  // once per rendered pixel, after the frame
  }
}

// ---- ObjLoader.swift -------------------------------------------------------------------
struct Vertex { double x, y, z; };
struct Face { int vertexIndices[3]; std::string materialName; };
struct ObjFile {
  std::vector<Triangle> triangles;  // createTriangleList output, material assigned by name
};

// readObjFile + readMtlFile + createTriangleList in one step (the C ABI keeps them together)
inline std::vector<Triangle> loadTriangles(const std::string& objFile, const std::string& mtlFile,
                                           std::map<std::string, Material>* materialDict = nullptr) {
  rtmi_obj_mesh* mesh = nullptr;
  int32_t rc = rtmi_obj_load(objFile.c_str(), mtlFile.empty() ? nullptr : mtlFile.c_str(), &mesh);
  if (rc != RTMI_OK) throw RtmiError(rc, "rtmi_obj_load");
  std::vector<Material> mats;
  const rtmi_material* cm = rtmi_obj_materials(mesh);
  for (int32_t i = 0; i < rtmi_obj_material_count(mesh); ++i) {
    Material m{cm[i].Ns, cm[i].d, cm[i].illum, RGB(cm[i].Kd[0], cm[i].Kd[1], cm[i].Kd[2]),
               RGB(cm[i].Ka[0], cm[i].Ka[1], cm[i].Ka[2]), RGB(cm[i].Ks[0], cm[i].Ks[1], cm[i].Ks[2])};
    mats.push_back(m);
    if (materialDict) materialDict->insert_or_assign(rtmi_obj_material_name(mesh, i), m);
  }
  std::vector<Triangle> out;
  const rtmi_triangle* t = rtmi_obj_triangles(mesh);
  for (int64_t i = 0; i < rtmi_obj_triangle_count(mesh); ++i) {
    std::vector<v3d> pts;
    for (int k = 0; k < 3; ++k) pts.emplace_back(t[i].v[k][0], t[i].v[k][1], t[i].v[k][2]);
    out.emplace_back(std::move(pts), t[i].material >= 0 ? std::optional<Material>(mats[(size_t)t[i].material]) : std::nullopt);
  }
  rtmi_obj_free(mesh);
  return out;
}

}  // namespace raytrace
