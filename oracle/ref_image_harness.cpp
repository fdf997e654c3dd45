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

int main(int argc, char** argv) {
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
