// oracle/ref_image_harness.cpp — TEST INFRASTRUCTURE ONLY (golden generation).
//
// Links the reference's own image writer — src/image.cpp + src/stb.cpp (vendored stb_image_write / stb_image and
// header-only GLM, nothing else) — compiled in place from /root/reference, fills an `image` exactly the way
// saveImage() does (src/main.cpp:91-97: img.setPixel(width - 1 - x, y, pix / samples)), calls image::savePNG
// (src/image.cpp:22-39) and reads the file back with the reference's own stb_image.  Prints the input SUM image
// (float bit patterns) and the bytes of the PNG as JSON: committed as tests/golden/ref_image.json, it pins the
// product's pt_save_png / pt_save_u8 (clamp, x 255, truncation, x mirror, NaN and out-of-range handling) to the
// reference's writer instead of to a re-derivation.
//
// Second mode (`ref_image hdr BASENAME OUT.json`): the same fill, then image::saveHDR (src/image.cpp:41-45, the call that
// main.cpp:106 keeps commented out) for a wide image (run-length path, runs longer than 127 and dumps longer than 128)
// and a narrow one (width < 8: flat RGBE); the files' bytes go into tests/golden/ref_hdr.json and pin pt_save_hdr.
//
// Built by `make -C oracle ref` into oracle/_ref/ (git-ignored); never shipped.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include <glm/glm.hpp>
#include <stb_image.h>

#include "image.h"

static int hdr_goldens(const std::string& base, FILE* out) {
  const float samples = 4.0f;
  fprintf(out, "{\n \"source\": \"reference src/image.cpp (saveHDR) + src/stb.cpp compiled in place; fill loop of src/main.cpp:91-97\",\n \"samples\": %g,\n \"cases\": [", samples);
  const int dims[2][2] = {{300, 3}, {5, 3}};
  for (int k = 0; k < 2; ++k) {
    const int width = dims[k][0], height = dims[k][1];
    std::vector<glm::vec3> sum(width * height);
    uint32_t s = 777u + k;
    auto rnd = [&]() {
      s = s * 1664525u + 1013904223u;
      return (float)(s >> 8) / 16777216.0f;
    };
    for (int y = 0; y < height; ++y)
      for (int x = 0; x < width; ++x) {
        glm::vec3 p;
        if (y == 0) p = x < 140 ? glm::vec3(2.0f, 0.5f, 0.125f) : x < 150 ? glm::vec3(rnd(), rnd(), rnd()) * 8.0f : glm::vec3(0.0f);  // long runs
        else if (y == 1) p = glm::vec3(rnd(), rnd(), rnd()) * (x % 7 == 0 ? 1e3f : 3.0f);                                         // long dumps
        else p = x % 50 < 3 ? glm::vec3(1e-33f, 2e-33f, 0.0f) : x % 50 < 6 ? glm::vec3(1e30f, 5e29f, 1.0f) : glm::vec3((float)(x / 10) * 0.25f);  // tiny, huge, short runs
        sum[x + y * width] = p * samples;
      }
    image img(width, height);
    for (int x = 0; x < width; x++)
      for (int y = 0; y < height; y++) img.setPixel(width - 1 - x, y, glm::vec3(sum[x + y * width]) / samples);
    const std::string b = base + (k ? "_narrow" : "_wide");
    img.saveHDR(b);
    FILE* f = fopen((b + ".hdr").c_str(), "rb");
    if (!f) return 1;
    std::vector<unsigned char> bytes;
    int ch;
    while ((ch = fgetc(f)) != EOF) bytes.push_back((unsigned char)ch);
    fclose(f);
    fprintf(out, "%s\n  {\"width\": %d, \"height\": %d, \"sum_bits\": [", k ? "," : "", width, height);
    for (int i = 0; i < width * height; ++i)
      for (int c = 0; c < 3; ++c) {
        uint32_t u;
        float v = sum[i][c];
        memcpy(&u, &v, 4);
        fprintf(out, "%u%s", u, (i == width * height - 1 && c == 2) ? "" : ", ");
      }
    fprintf(out, "],\n   \"hdr_file_hex\": \"");
    for (unsigned char c : bytes) fprintf(out, "%02x", c);
    fprintf(out, "\"}");
  }
  fprintf(out, "\n ]\n}\n");
  return 0;
}

int main(int argc, char** argv) {
  if (argc > 1 && std::string(argv[1]) == "hdr") {
    FILE* o = argc > 3 ? fopen(argv[3], "w") : stdout;
    if (!o) return 1;
    const int rc = hdr_goldens(argc > 2 ? argv[2] : "/tmp/ref_hdr_golden", o);
    if (o != stdout) fclose(o);
    return rc;
  }
  // usage: ref_image PNG_BASENAME OUT.json   (image::savePNG itself prints "Saved ..." on stdout)
  const std::string base = argc > 1 ? argv[1] : "/tmp/ref_image_golden";
  FILE* out = argc > 2 ? fopen(argv[2], "w") : stdout;
  if (!out) return 1;
  const int width = 9, height = 4;
  const float samples = 3.0f;
  std::vector<glm::vec3> sum(width * height);
  uint32_t s = 12345u;
  auto rnd = [&]() {
    s = s * 1664525u + 1013904223u;
    return (float)(s >> 8) / 16777216.0f;
  };
  for (auto& p : sum) p = glm::vec3(rnd(), rnd(), rnd()) * samples;
  // special values (as running sums over `samples` iterations)
  sum[0] = glm::vec3(-1.0f, 0.0f, -0.0f);                                     // negative, zeros
  sum[1] = glm::vec3(3.0f, 3.0000002f, 2.9999998f);                           // exactly 1, just above, just below
  sum[2] = glm::vec3(7.5f, 1e30f, INFINITY);                                  // far above 1
  sum[3] = glm::vec3(NAN, 1.5f, 0.75f);                                       // NaN channel
  sum[4] = glm::vec3(3.0f / 255.0f, 2.999f / 255.0f, 3.001f / 255.0f);        // around a byte boundary
  sum[5] = glm::vec3(254.5f / 255.0f * 3.0f, 254.999f / 255.0f * 3.0f, 1e-30f);
  sum[6] = glm::vec3(-INFINITY, 1.5e-45f, 0.5f * 3.0f);
  image img(width, height);
  for (int x = 0; x < width; x++)
    for (int y = 0; y < height; y++) {
      int index = x + (y * width);
      glm::vec3 pix = sum[index];
      img.setPixel(width - 1 - x, y, glm::vec3(pix) / samples);
    }
  img.savePNG(base);
  int w = 0, h = 0, n = 0;
  unsigned char* px = stbi_load((base + ".png").c_str(), &w, &h, &n, 3);
  if (!px || w != width || h != height) {
    fprintf(stderr, "cannot read %s.png back\n", base.c_str());
    return 1;
  }
  fprintf(stderr, "read back %dx%d, %d channels in file\n", w, h, n);
  fprintf(out, "{\n \"source\": \"reference src/image.cpp + src/stb.cpp compiled in place; fill loop of src/main.cpp:91-97\",\n");
  fprintf(out, " \"width\": %d, \"height\": %d, \"samples\": %g, \"file_channels\": %d,\n \"sum_bits\": [", width, height, samples, n);
  for (int i = 0; i < width * height; ++i)
    for (int c = 0; c < 3; ++c) {
      uint32_t u;
      float f = sum[i][c];
      memcpy(&u, &f, 4);
      fprintf(out, "%u%s", u, (i == width * height - 1 && c == 2) ? "" : ", ");
    }
  fprintf(out, "],\n \"png_rgb8\": [");
  for (int i = 0; i < w * h * 3; ++i) fprintf(out, "%d%s", px[i], i == w * h * 3 - 1 ? "" : ", ");
  fprintf(out, "]\n}\n");
  if (out != stdout) fclose(out);
  stbi_image_free(px);
  return 0;
}
