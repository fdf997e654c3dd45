// oracle/ref_xform_harness.cpp — TEST INFRASTRUCTURE ONLY (golden generation).
//
// The first of the reference-compiled harnesses (round 1): src/utilities.cpp + the vendored header-only GLM 0.9.6.3
// need no CUDA header at all.  (scene.cpp / intersections.h reach <cuda_runtime.h>; a genuine one ships inside this
// image's triton package and ref_hot_harness.cpp compiles them against it.  The image writer, src/image.cpp +
// src/stb.cpp: ref_image_harness.cpp.)  This harness links the reference's own
// utilityCore::buildTransformationMatrix (src/utilities.cpp:64-72) and calls GLM's
// inverse / inverseTranspose exactly as scene.cpp:83-86 does, plus the handful of
// GLM vector primitives the hot path uses, and prints the results as JSON.  The
// output is committed as tests/golden/ref_xforms.json and pins the oracle's
// (and the product's) GLM restatement bit-for-bit.
//
// Built by `make -C oracle ref` into oracle/_ref/ (git-ignored); never shipped.
#include <cstdio>
#include <cstring>
#include <cstdint>
#include <glm/glm.hpp>
#include <glm/gtc/matrix_inverse.hpp>
#include "utilities.h"

static void pm(const char* name, const glm::mat4& m, bool last) {
  printf("    \"%s\": [", name);
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) {
      uint32_t u;
      float f = m[c][r];
      memcpy(&u, &f, 4);
      printf("%u%s", u, (c == 3 && r == 3) ? "" : ", ");
    }
  printf("]%s\n", last ? "" : ",");
}
static uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

int main() {
  // TRANS / ROTAT / SCALE triples: the seven cornell.txt objects, sphere.txt's, and a few
  // generic rotations (all three axes non-zero) to exercise every term.
  const float T[][9] = {
      {0, 10, 0, 0, 0, 0, 3, .3f, 3},      {0, 0, 0, 0, 0, 0, 10, .01f, 10},  {0, 10, 0, 0, 0, 90, .01f, 10, 10},
      {0, 5, -5, 0, 90, 0, .01f, 10, 10},  {-5, 5, 0, 0, 0, 0, .01f, 10, 10}, {5, 5, 0, 0, 0, 0, .01f, 10, 10},
      {-1, 4, -1, 0, 0, 0, 3, 3, 3},       {0, 0, 0, 0, 0, 0, 3, 3, 3},       {1.25f, -2.5f, 3.75f, 30, 45, 60, 1, 2, 3},
      {-0.4f, 0.6f, 2.2f, 0, 63, 0, .25f, .25f, .25f}, {2, 1, -3, 17.5f, -80, 123, 0.5f, 4, 0.125f}};
  int n = sizeof(T) / sizeof(T[0]);
  printf("{\n  \"xforms\": [\n");
  for (int i = 0; i < n; ++i) {
    glm::vec3 t((double)T[i][0], (double)T[i][1], (double)T[i][2]);
    glm::vec3 r((double)T[i][3], (double)T[i][4], (double)T[i][5]);
    glm::vec3 s((double)T[i][6], (double)T[i][7], (double)T[i][8]);
    glm::mat4 M = utilityCore::buildTransformationMatrix(t, r, s);
    glm::mat4 I = glm::inverse(M);
    glm::mat4 IT = glm::inverseTranspose(M);
    printf("   {\"trs\": [%u, %u, %u, %u, %u, %u, %u, %u, %u],\n", bits(t.x), bits(t.y), bits(t.z), bits(r.x), bits(r.y),
           bits(r.z), bits(s.x), bits(s.y), bits(s.z));
    pm("transform", M, false);
    pm("inverse", I, false);
    pm("invTranspose", IT, true);
    printf("   }%s\n", i + 1 < n ? "," : "");
  }
  printf("  ],\n  \"vecops\": [\n");
  // normalize / cross / dot / length / mat4*vec4 on a few vectors
  const float V[][6] = {{1, 2, 3, -4, 5, 0.5f}, {0.1f, -0.2f, 0.3f, 7, 11, -13}, {0, 4.37113883e-08f, -1, 0, 1, 0},
                        {1e-3f, 2e5f, -3.3f, 0.577f, 0.577f, 0.577f}};
  int nv = sizeof(V) / sizeof(V[0]);
  glm::mat4 M = utilityCore::buildTransformationMatrix(glm::vec3(1.25f, -2.5f, 3.75f), glm::vec3(30, 45, 60), glm::vec3(1, 2, 3));
  for (int i = 0; i < nv; ++i) {
    glm::vec3 a(V[i][0], V[i][1], V[i][2]), b(V[i][3], V[i][4], V[i][5]);
    glm::vec3 na = glm::normalize(a), cr = glm::cross(a, b);
    glm::vec3 mp = glm::vec3(M * glm::vec4(a, 1.0f)), md = glm::vec3(M * glm::vec4(a, 0.0f));
    printf("   {\"a\": [%u, %u, %u], \"b\": [%u, %u, %u], \"normalize_a\": [%u, %u, %u], \"cross\": [%u, %u, %u], "
           "\"dot\": %u, \"length_a\": %u, \"M_point\": [%u, %u, %u], \"M_dir\": [%u, %u, %u]}%s\n",
           bits(a.x), bits(a.y), bits(a.z), bits(b.x), bits(b.y), bits(b.z), bits(na.x), bits(na.y), bits(na.z),
           bits(cr.x), bits(cr.y), bits(cr.z), bits(glm::dot(a, b)), bits(glm::length(a)), bits(mp.x), bits(mp.y),
           bits(mp.z), bits(md.x), bits(md.y), bits(md.z), i + 1 < nv ? "," : "");
  }
  printf("  ]\n}\n");
  return 0;
}
