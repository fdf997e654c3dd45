// pt_image.cpp — image output of the MI355X path tracer: what the reference's
// saveImage() (src/main.cpp:86-107) + image::savePNG (src/image.cpp:22-39) produce,
// without stb: a self-contained PNG encoder (zlib "stored" deflate blocks) and a
// PFM writer for lossless float output.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <algorithm>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/pt_amd.h"

namespace {

uint32_t crc_table[256];
bool crc_ready = false;
uint32_t crc32(const uint8_t* p, size_t n, uint32_t c = 0xffffffffu) {
  if (!crc_ready) {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t v = i;
      for (int k = 0; k < 8; ++k) v = (v & 1) ? 0xedb88320u ^ (v >> 1) : v >> 1;
      crc_table[i] = v;
    }
    crc_ready = true;
  }
  for (size_t i = 0; i < n; ++i) c = crc_table[(c ^ p[i]) & 0xff] ^ (c >> 8);
  return c;
}
void be32(std::vector<uint8_t>& v, uint32_t x) {
  v.push_back(x >> 24), v.push_back(x >> 16), v.push_back(x >> 8), v.push_back(x);
}
void chunk(std::vector<uint8_t>& out, const char* type, const std::vector<uint8_t>& data) {
  be32(out, (uint32_t)data.size());
  std::vector<uint8_t> td(type, type + 4);
  td.insert(td.end(), data.begin(), data.end());
  out.insert(out.end(), td.begin(), td.end());
  be32(out, crc32(td.data(), td.size()) ^ 0xffffffffu);
}
// zlib stream of stored (uncompressed) deflate blocks
std::vector<uint8_t> zstore(const std::vector<uint8_t>& raw) {
  std::vector<uint8_t> z = {0x78, 0x01};
  size_t pos = 0;
  do {
    const size_t n = std::min<size_t>(65535, raw.size() - pos);
    const bool last = pos + n == raw.size();
    z.push_back(last ? 1 : 0);
    z.push_back(n & 0xff), z.push_back(n >> 8);
    z.push_back(~n & 0xff), z.push_back((~n >> 8) & 0xff);
    z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
    pos += n;
  } while (pos < raw.size());
  uint32_t a = 1, b = 0;
  for (uint8_t c : raw) {
    a = (a + c) % 65521u;
    b = (b + a) % 65521u;
  }
  be32(z, (b << 16) | a);
  return z;
}
// clamp(pix, 0, 1) * 255 truncated to u8, as image.cpp:26-30 (glm::clamp = min(max(x,0),1))
uint8_t to_u8(float v) {
  float m = v > 0.0f ? v : 0.0f;
  m = m < 1.0f ? m : 1.0f;
  return (uint8_t)(m * 255.f);
}

}  // namespace

extern "C" {

int pt_write_png_rgb8(const char* path, const uint8_t* rgb8, int w, int h) {
  if (!path || !rgb8 || w <= 0 || h <= 0) return -1;
  std::vector<uint8_t> raw;
  raw.reserve((size_t)h * (3 * (size_t)w + 1));
  for (int y = 0; y < h; ++y) {
    raw.push_back(0);  // filter: none
    raw.insert(raw.end(), rgb8 + (size_t)y * 3 * w, rgb8 + (size_t)(y + 1) * 3 * w);
  }
  std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  std::vector<uint8_t> ihdr;
  be32(ihdr, (uint32_t)w), be32(ihdr, (uint32_t)h);
  ihdr.push_back(8), ihdr.push_back(2), ihdr.push_back(0), ihdr.push_back(0), ihdr.push_back(0);
  chunk(out, "IHDR", ihdr);
  chunk(out, "IDAT", zstore(raw));
  chunk(out, "IEND", {});
  FILE* f = fopen(path, "wb");
  if (!f) return -1;
  const size_t n = fwrite(out.data(), 1, out.size(), f);
  fclose(f);
  return n == out.size() ? 0 : -1;
}

int pt_save_png(const char* path, const float* rgb_sum, int w, int h, float samples) {
  if (!path || !rgb_sum || w <= 0 || h <= 0) return -1;
  std::vector<uint8_t> bytes((size_t)w * h * 3);
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      // saveImage(): img.setPixel(width - 1 - x, y, pix / samples) — output column x shows source column w-1-x
      const float* s = rgb_sum + 3 * ((size_t)(w - 1 - x) + (size_t)y * w);
      uint8_t* d = bytes.data() + 3 * ((size_t)x + (size_t)y * w);
      d[0] = to_u8(s[0] / samples), d[1] = to_u8(s[1] / samples), d[2] = to_u8(s[2] / samples);
    }
  return pt_write_png_rgb8(path, bytes.data(), w, h);
}

int pt_output_basename(const char* name, int samples, char* out, int cap) {
  static std::string start_time;  // main.cpp:35: taken once, at program start
  if (start_time.empty()) {
    time_t now;
    time(&now);
    char buf[sizeof "0000-00-00_00-00-00z"];
    strftime(buf, sizeof buf, "%Y-%m-%d_%H-%M-%Sz", gmtime(&now));
    start_time = buf;
  }
  std::ostringstream ss;
  ss << (name ? name : "") << "." << start_time << "." << (float)samples << "samp";
  const std::string s = ss.str();
  if (out && cap > 0) {
    const size_t n = std::min<size_t>(s.size(), (size_t)cap - 1);
    std::memcpy(out, s.data(), n);
    out[n] = 0;
  }
  return (int)s.size();
}

}  // extern "C"

// Radiance RGBE (.hdr) as image::saveHDR writes it through stb_image_write (src/image.cpp:41-45; the call saveImage()
// keeps commented out, main.cpp:106), with saveImage()'s x mirror and division by the sample count.  Byte-identical to
// the reference-compiled writer on tests/golden/ref_hdr.json.  The format: a text header, then per scanline either flat
// RGBE quadruples (width < 8 or >= 32768) or the marker {2, 2, width hi, width lo} followed by the four byte planes
// (R, G, B, E), each run-length coded: (128 + n, value) for n <= 127 equal bytes — only runs of at least three are
// coded that way — and (n, n literal bytes) for n <= 128 others.
namespace {
void to_rgbe(const float* rgb, uint8_t out[4]) {
  const float top = std::max(rgb[0], std::max(rgb[1], rgb[2]));
  if (top < 1e-32) {
    out[0] = out[1] = out[2] = out[3] = 0;
    return;
  }
  int e = 0;
  const float scale = (float)frexp(top, &e) * 256.0f / top;  // mantissa of the largest channel lands in [128, 256)
  out[0] = (uint8_t)(rgb[0] * scale), out[1] = (uint8_t)(rgb[1] * scale), out[2] = (uint8_t)(rgb[2] * scale);
  out[3] = (uint8_t)(e + 128);
}
void rle_plane(const uint8_t* v, int n, std::vector<uint8_t>& out) {
  int at = 0;
  while (at < n) {
    int run = at;  // start of the next run of >= 3 equal bytes, or n when there is none
    while (run + 2 < n && !(v[run] == v[run + 1] && v[run] == v[run + 2])) ++run;
    const bool found = run + 2 < n;
    if (!found) run = n;
    for (; at < run;) {  // literals up to the run
      const int len = std::min(run - at, 128);
      out.push_back((uint8_t)len);
      out.insert(out.end(), v + at, v + at + len);
      at += len;
    }
    if (found) {
      int end = run;
      while (end < n && v[end] == v[at]) ++end;
      for (; at < end;) {
        const int len = std::min(end - at, 127);
        out.push_back((uint8_t)(128 + len));
        out.push_back(v[at]);
        at += len;
      }
    }
  }
}
}  // namespace

extern "C" int pt_save_hdr(const char* path, const float* rgb_sum, int w, int h, float samples) {
  if (!path || !rgb_sum || w <= 0 || h <= 0) return -1;
  std::vector<uint8_t> out;
  char head[160];
  const int hn = snprintf(head, sizeof head,
                          "#?RADIANCE\n# Written by stb_image_write.h\nFORMAT=32-bit_rle_rgbe\nEXPOSURE=          1.0000000000000\n\n-Y %d +X %d\n", h, w);
  out.insert(out.end(), head, head + hn);
  std::vector<uint8_t> planes(4 * (size_t)w);
  for (int y = 0; y < h; ++y) {
    for (int x = 0; x < w; ++x) {
      const float* s = rgb_sum + 3 * ((size_t)(w - 1 - x) + (size_t)y * w);  // saveImage(): setPixel(width - 1 - x, y, pix / samples)
      const float px[3] = {s[0] / samples, s[1] / samples, s[2] / samples};
      uint8_t q[4];
      to_rgbe(px, q);
      for (int c = 0; c < 4; ++c) planes[(size_t)c * w + x] = q[c];
    }
    if (w < 8 || w >= 32768) {
      for (int x = 0; x < w; ++x)
        for (int c = 0; c < 4; ++c) out.push_back(planes[(size_t)c * w + x]);
    } else {
      out.push_back(2), out.push_back(2), out.push_back((uint8_t)((w >> 8) & 0xff)), out.push_back((uint8_t)(w & 0xff));
      for (int c = 0; c < 4; ++c) rle_plane(planes.data() + (size_t)c * w, w, out);
    }
  }
  FILE* f = fopen(path, "wb");
  if (!f) return -1;
  const bool ok = fwrite(out.data(), 1, out.size(), f) == out.size();
  fclose(f);
  return ok ? 0 : -1;
}

extern "C" {
// Little-endian PFM, rows bottom-to-top per the format; raw orientation (no x mirror),
// averaged radiance — the loss-free companion to the PNG for PSNR work.
int pt_save_pfm(const char* path, const float* rgb_sum, int w, int h, float samples) {
  if (!path || !rgb_sum || w <= 0 || h <= 0) return -1;
  FILE* f = fopen(path, "wb");
  if (!f) return -1;
  fprintf(f, "PF\n%d %d\n-1.0\n", w, h);
  std::vector<float> row(3 * (size_t)w);
  for (int y = h - 1; y >= 0; --y) {
    for (size_t i = 0; i < row.size(); ++i) row[i] = rgb_sum[(size_t)y * w * 3 + i] / samples;
    if (fwrite(row.data(), sizeof(float), row.size(), f) != row.size()) {
      fclose(f);
      return -1;
    }
  }
  fclose(f);
  return 0;
}

}  // extern "C"
