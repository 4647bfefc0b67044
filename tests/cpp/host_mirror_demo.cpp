// Drives librtmi.so through the C++ mirror of the reference's Swift API, the way startRT does
// (RayTraceUI/ViewController.swift:137-180), and writes the bitmaps for the Python test to compare
// with the oracle.  Usage: host_mirror_demo <out_dir> <cube.obj> <cube.mtl>
#include <cstdio>
#include <fstream>
#include <string>

#include "RayTraceHost.hpp"

using namespace raytrace;

static void write_file(const std::string& path, const std::vector<uint8_t>& data) {
  std::ofstream f(path, std::ios::binary);
  f.write(reinterpret_cast<const char*>(data.data()), (std::streamsize)data.size());
}

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  const std::string out = argv[1];
  try {
    // --- the hard-coded scene of startRT ------------------------------------------------
    imageWidth = 320;
    imageHeight = 240;
    std::vector<uint8_t> outputBitmap((size_t)4 * imageWidth * imageHeight, 0);
    std::vector<std::shared_ptr<PointLight>> lights = {
        std::make_shared<PointLight>(v3d(50, 0, 50), RGB(0.8, 0.8, 0.8), RGB(0.5, 0.5, 0.5)),
        std::make_shared<PointLight>(v3d(-50, 0.0, -50), RGB(0.8, 0.8, 0.8), RGB(0.5, 0.5, 0.5))};
    std::map<std::string, Material> materialDict;
    Material m{8, 10, 5, RGB(0.5, 0.5, 0.5), RGB(0.7, 0.7, 0.7), RGB(1.0, 1.0, 1.0)};
    Sphere s(v3d(25, 0, 0), 25);
    s.material = m;
    Sphere s1(v3d(-25, 0, 0), 25);
    s1.material = m;
    int pixelCounter = 0;
    raytraceWorld(v3d(0, 0, 40), v3d(0, 0, -1), 1.0, imageWidth - 1, imageHeight - 1, RGB(0.2, 0.2, 0.2), lights,
                  {s, s1}, materialDict, outputBitmap, [&] { pixelCounter += 1; });
    if (pixelCounter != (imageWidth - 1) * (imageHeight - 1)) { fprintf(stderr, "pixelDone count %d\n", pixelCounter); return 1; }
    write_file(out + "/two_sphere.rgba", outputBitmap);

    // --- loadObjFile: OBJ + MTL -> [Triangle] -> Sphere(boundingObjects:) ----------------
    imageWidth = 128;
    imageHeight = 128;
    std::vector<uint8_t> bitmap2((size_t)4 * imageWidth * imageHeight, 0);
    std::vector<Triangle> triangles = loadTriangles(argv[2], argv[3], &materialDict);
    Sphere bound(triangles);
    Triangle floor({v3d(-3, -0.8, -3), v3d(3, -0.8, 3), v3d(3, -0.8, -3)}, m);
    raytraceWorld(v3d(1.2, 0.9, 3.0), v3d(-0.35, -0.25, -1.0), 8.0, 0, 0, RGB(0.2, 0.2, 0.2), lights,
                  {bound, floor, lights[1], s1}, materialDict, bitmap2, nullptr, 0.5);
    write_file(out + "/cube.rgba", bitmap2);
    printf("bounding sphere centre %.17g %.17g %.17g radius %.17g\n", bound.location.x, bound.location.y, bound.location.z,
           bound.radius);
    // error behaviour: a bitmap of the wrong size is rejected before anything is rendered
    std::vector<uint8_t> tooSmall(16);
    try {
      raytraceWorld(v3d(0, 0, 40), v3d(0, 0, -1), 1.0, 0, 0, RGB(0.2, 0.2, 0.2), lights, {s}, materialDict, tooSmall);
      return 1;
    } catch (const std::out_of_range&) {
    }
  } catch (const std::exception& e) {
    fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
