// rtmi_image.cpp -- PPM and PNG writers for the RGBA8 frame (host only).
//
// Stands in for the reference's on-screen path: the CGImage built from outputBitmap with
// bytesPerRow = 4*width, 8 bits per component, alpha last (ViewController.swift:197-216).
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/rtmi.h"

void rtmi_set_last_error(const std::string& msg);  // rtmi.hip

namespace {

struct File {
  FILE* f;
  explicit File(const char* path) : f(fopen(path, "wb")) {}
  ~File() { if (f) fclose(f); }
};

uint32_t crc32_update(uint32_t crc, const uint8_t* p, size_t n) {
  static uint32_t table[256];
  static bool ready = false;
  if (!ready) {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
      table[i] = c;
    }
    ready = true;
  }
  for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xff] ^ (crc >> 8);
  return crc;
}

void put_be32(std::vector<uint8_t>& v, uint32_t x) {
  v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x);
}

bool write_chunk(FILE* f, const char type[4], const std::vector<uint8_t>& data) {
  std::vector<uint8_t> head;
  put_be32(head, (uint32_t)data.size());
  head.insert(head.end(), type, type + 4);
  uint32_t crc = crc32_update(0xffffffffu, head.data() + 4, 4);
  crc = crc32_update(crc, data.data(), data.size()) ^ 0xffffffffu;
  std::vector<uint8_t> tail;
  put_be32(tail, crc);
  return fwrite(head.data(), 1, head.size(), f) == head.size() &&
         (data.empty() || fwrite(data.data(), 1, data.size(), f) == data.size()) &&
         fwrite(tail.data(), 1, tail.size(), f) == tail.size();
}

int32_t check(const char* path, const uint8_t* rgba, int32_t width, int32_t height) {
  if (!path || !rgba || width <= 0 || height <= 0) return RTMI_ERR_INVALID_ARGUMENT;
  return RTMI_OK;
}

}  // namespace

// Binary PPM (P6): the RGB bytes of every pixel, alpha dropped.
extern "C" int32_t rtmi_write_ppm(const char* path, const uint8_t* rgba, int32_t width, int32_t height) {
  int32_t rc = check(path, rgba, width, height);
  if (rc) return rc;
  File out(path);
  if (!out.f) { rtmi_set_last_error(std::string("cannot open ") + path); return RTMI_ERR_INVALID_ARGUMENT; }
  fprintf(out.f, "P6\n%d %d\n255\n", width, height);
  std::vector<uint8_t> row((size_t)width * 3);
  for (int32_t y = 0; y < height; ++y) {
    const uint8_t* src = rgba + (size_t)y * width * 4;
    for (int32_t x = 0; x < width; ++x) { row[3 * x] = src[4 * x]; row[3 * x + 1] = src[4 * x + 1]; row[3 * x + 2] = src[4 * x + 2]; }
    if (fwrite(row.data(), 1, row.size(), out.f) != row.size()) { rtmi_set_last_error(std::string("short write to ") + path); return RTMI_ERR_INVALID_ARGUMENT; }
  }
  return RTMI_OK;
}

// PNG, 8-bit RGBA, filter 0 on every row, zlib stream of stored (uncompressed) deflate blocks:
// every PNG reader accepts it and no compression library is needed.
extern "C" int32_t rtmi_write_png(const char* path, const uint8_t* rgba, int32_t width, int32_t height) {
  int32_t rc = check(path, rgba, width, height);
  if (rc) return rc;
  File out(path);
  if (!out.f) { rtmi_set_last_error(std::string("cannot open ") + path); return RTMI_ERR_INVALID_ARGUMENT; }
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  bool ok = fwrite(sig, 1, 8, out.f) == 8;
  std::vector<uint8_t> ihdr;
  put_be32(ihdr, (uint32_t)width);
  put_be32(ihdr, (uint32_t)height);
  ihdr.push_back(8); ihdr.push_back(6); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);  // 8-bit RGBA, no interlace
  ok = ok && write_chunk(out.f, "IHDR", ihdr);

  const size_t stride = (size_t)width * 4 + 1;  // filter byte + row
  std::vector<uint8_t> raw(stride * (size_t)height);
  for (int32_t y = 0; y < height; ++y) {
    raw[(size_t)y * stride] = 0;
    std::copy(rgba + (size_t)y * width * 4, rgba + (size_t)(y + 1) * width * 4, raw.begin() + (size_t)y * stride + 1);
  }
  uint32_t a = 1, b = 0;  // Adler-32 of the raw stream
  for (size_t i = 0; i < raw.size();) {
    size_t n = std::min<size_t>(5552, raw.size() - i);
    for (size_t k = 0; k < n; ++k) { a += raw[i + k]; b += a; }
    a %= 65521u; b %= 65521u;
    i += n;
  }
  std::vector<uint8_t> z;
  z.reserve(raw.size() + raw.size() / 65535 * 5 + 16);
  z.push_back(0x78); z.push_back(0x01);
  size_t pos = 0;
  do {
    size_t n = std::min<size_t>(65535, raw.size() - pos);
    z.push_back(pos + n == raw.size() ? 1 : 0);
    z.push_back((uint8_t)(n & 0xff)); z.push_back((uint8_t)(n >> 8));
    z.push_back((uint8_t)(~n & 0xff)); z.push_back((uint8_t)((~n >> 8) & 0xff));
    z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
    pos += n;
  } while (pos < raw.size());
  put_be32(z, (b << 16) | a);
  ok = ok && write_chunk(out.f, "IDAT", z) && write_chunk(out.f, "IEND", {});
  if (!ok) { rtmi_set_last_error(std::string("short write to ") + path); return RTMI_ERR_INVALID_ARGUMENT; }
  return RTMI_OK;
}
