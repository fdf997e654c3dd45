// oracle/ref_hot_harness.cpp — TEST INFRASTRUCTURE ONLY (golden generation).
//
// Pins the hot path to the reference's OWN code, compiled in place from /root/reference:
//   * src/intersections.h (included unmodified): utilhash :12-20, getPointOnRay :27-29, multiplyMV :34-36,
//     boxIntersectionTest :48-90, sphereIntersectionTest :102-144;
//   * src/scene.cpp (linked): Scene::Scene :7-33, loadGeom :35-90, loadCamera :92-151, loadMaterial :153-188;
//   * src/utilities.cpp (linked): buildTransformationMatrix :64-72, tokenizeString, safeGetline :88-112;
//   * the vendored header-only GLM 0.9.6.3.
// The only CUDA dependency of those files is `#include <cuda_runtime.h>` (sceneStructs.h:5).  A genuine NVIDIA
// cuda_runtime.h ships in this image inside the installed triton package (triton/backends/nvidia/include); the Makefile
// locates it at build time and skips this harness where it is absent.  No stand-in header is written.  Two lines of
// glue are needed because CUDA's global min()/max() exist only under nvcc: `using std::min; using std::max;` before
// the include.  `new Scene(path)` is used exactly as main.cpp:45 does, so the undefined Scene::~Scene (scene.h:21) is
// never needed.
//
// What of src/pathtrace.cu is pinned where: its plain-C++ functions (buildBVH / computeBounds, intersectAABB, the sampling
// helpers) are compiled from the file's own text by ref_pt_harness.cpp; its __global__ bodies (computeIntersections' stack
// walk, shadeAndExtendRays, finalGather) and main.cpp's camera fix-up compile with neither g++ nor clang's CUDA mode without
// stand-ins for the CUDA toolchain and stay pinned by restatement + SURVEY §4 KATs.  The one formula of pathtrace.cu this
// harness restates is the seed expression of makeSeededRandomEngine (pathtrace.cu:205) — around the reference's own
// utilhash — to produce the seed list that ref_rng_harness.cpp feeds to rocThrust's minstd_rand.
//
// Modes:
//   ref_hot scene OUT.json SCENE.txt...        dump geoms / materials / camera / render state as the reference's loader
//                                              leaves them (bit patterns)
//   ref_hot isect OUT.bin  SCENE.txt...        per scene: a deterministic ray set (camera-like, random, origins inside
//                                              primitives, axis-parallel, grazing/corner, bounce chains) tested against
//                                              EVERY geom with the reference's box / sphere test; plus utilhash vectors
//                                              and the seed list
// Built by `make -C oracle ref` into oracle/_ref/ (git-ignored); never shipped; only the fixtures travel.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
using std::max;  // CUDA's global min()/max() overloads exist only under nvcc; intersections.h:131,134 call them unqualified
using std::min;

#include "intersections.h"
#include "scene.h"

#include "ref_gold_io.h"

using gold::fbits;

// ───────────────────────── deterministic inputs (our own generator; inputs only) ─────────────────────────
struct Rng {
  uint32_t s;
  explicit Rng(uint32_t seed) : s(seed ? seed : 1u) {}
  uint32_t next() {
    s ^= s << 13;
    s ^= s >> 17;
    s ^= s << 5;
    return s;
  }
  float u() { return (float)(next() >> 8) * (1.0f / 16777216.0f); }  // [0,1)
  float sym() { return 2.0f * u() - 1.0f; }                           // [-1,1)
  glm::vec3 unit() {
    for (;;) {
      glm::vec3 v(sym(), sym(), sym());
      float l2 = glm::dot(v, v);
      if (l2 > 1e-3f && l2 <= 1.0f) return glm::normalize(v);
    }
  }
};

static float geomTest(const Geom& g, const Ray& r, glm::vec3& p, glm::vec3& n, bool& outside) {
  // dispatch of computeIntersections, pathtrace.cu:311-313 (CUBE → box test, otherwise sphere test)
  return g.type == CUBE ? boxIntersectionTest(g, r, p, n, outside) : sphereIntersectionTest(g, r, p, n, outside);
}

static void pushRay(std::vector<Ray>& rays, const glm::vec3& o, const glm::vec3& d) {
  Ray r;
  r.origin = o;
  r.direction = d;
  rays.push_back(r);
}

// The ray set of one scene.  Counts are fixed so the golden's size is known: 1536 rays.
static std::vector<Ray> makeRays(const std::vector<Geom>& geoms, uint32_t seed) {
  std::vector<Ray> rays;
  Rng rng(seed);
  const int G = (int)geoms.size();
  // A: 256 camera-like rays from cornell.txt's eye; the 16x16 lattice contains the exact image diagonals
  //    (x == ±y), which look along the box's corner edges where two walls tie in t
  const glm::vec3 eye(0.0f, 5.0f, 10.5f);
  for (int j = 0; j < 16; ++j)
    for (int i = 0; i < 16; ++i) {
      float x = (float)(i - 8) * 0.125f + 0.0625f * (float)((i + j) & 1);
      float y = (float)(j - 8) * 0.125f + 0.0625f * (float)((i + j) & 1);
      pushRay(rays, eye, glm::normalize(glm::vec3(x, y, -1.0f)));
    }
  // B: 256 random rays through the scene's neighbourhood, three of four aimed at a random point of a primitive's
  //    object-space cube (so that they hit something; the rest go anywhere)
  for (int k = 0; k < 256; ++k) {
    glm::vec3 o(9.0f * rng.sym(), 5.0f + 6.0f * rng.sym(), 1.0f + 10.0f * rng.sym());
    glm::vec3 d = rng.unit();
    if (k % 4) {
      glm::vec3 q(0.5f * rng.sym(), 0.5f * rng.sym(), 0.5f * rng.sym());
      glm::vec3 tgt = glm::vec3(geoms[k % G].transform * glm::vec4(q, 1.0f));
      if (glm::dot(tgt - o, tgt - o) > 1e-6f) d = glm::normalize(tgt - o);
    }
    pushRay(rays, o, d);
  }
  // C: 192 origins inside a primitive's object-space cube (inside the sphere or between sphere and its cube)
  for (int k = 0; k < 192; ++k) {
    const Geom& g = geoms[k % G];
    glm::vec3 q(0.45f * rng.sym(), 0.45f * rng.sym(), 0.45f * rng.sym());
    glm::vec3 o = glm::vec3(g.transform * glm::vec4(q, 1.0f));
    pushRay(rays, o, rng.unit());
  }
  // D: 192 rays with zero direction components (0 * inf and x / 0 in the slab arithmetic), some origins on planes
  for (int k = 0; k < 192; ++k) {
    glm::vec3 o(6.0f * rng.sym(), 5.0f + 6.0f * rng.sym(), 6.0f * rng.sym());
    if (k % 4 == 1) o.y = 10.0f;   // cornell's ceiling plane
    if (k % 4 == 2) o.x = -5.0f;   // left wall plane
    if (k % 8 == 3) o = glm::vec3(geoms[k % G].transform * glm::vec4(0.0f, 0.0f, 0.0f, 1.0f));  // a primitive's centre
    glm::vec3 d(0.0f);
    int axis = k % 3;
    if (k % 2 == 0) {
      d[axis] = (k & 8) ? 1.0f : -1.0f;  // axis-parallel
    } else {
      d = rng.unit();
      d[axis] = 0.0f;  // one zero component
      d = glm::normalize(d);
    }
    pushRay(rays, o, d);
  }
  // E: 256 grazing rays: aimed exactly at a primitive's transformed corner / edge midpoint / face centre, and
  //    tangents of the unit sphere
  for (int k = 0; k < 256; ++k) {
    const Geom& g = geoms[k % G];
    int code = (int)(rng.next() % 27u);
    if (code == 13) code = 0;  // skip the centre
    glm::vec3 q(0.5f * (float)(code % 3 - 1), 0.5f * (float)((code / 3) % 3 - 1), 0.5f * (float)(code / 9 - 1));
    if (g.type == SPHERE && (k & 1)) {
      glm::vec3 s = 0.5f * rng.unit();             // a point on the unit sphere (object space)
      glm::vec3 t = glm::normalize(glm::cross(s, rng.unit()));
      glm::vec3 oo = s + 3.0f * t;                 // on the tangent line
      glm::vec3 o = glm::vec3(g.transform * glm::vec4(oo, 1.0f));
      glm::vec3 tgt = glm::vec3(g.transform * glm::vec4(s, 1.0f));
      pushRay(rays, o, glm::normalize(tgt - o));
      continue;
    }
    glm::vec3 tgt = glm::vec3(g.transform * glm::vec4(q, 1.0f));
    glm::vec3 o(8.0f * rng.sym(), 5.0f + 7.0f * rng.sym(), 8.0f * rng.sym());
    glm::vec3 d = tgt - o;
    if (glm::dot(d, d) < 1e-6f) d = glm::vec3(0.0f, 0.0f, -1.0f);
    pushRay(rays, o, glm::normalize(d));
  }
  // F: 384 rays of bounce chains (3 depths x 128): origin = hit point + n * 1e-3 as shadeAndExtendRays leaves it
  //    (pathtrace.cu:419,431), direction alternately the mirror direction (not re-normalised, as :240) and a
  //    random direction about the normal
  std::vector<Ray> cur;
  for (int k = 0; k < 64; ++k) cur.push_back(rays[2 * k + (k & 1)]);       // from A
  for (int k = 0; k < 64; ++k) cur.push_back(rays[256 + k]);               // from B
  for (int depth = 1; depth <= 3; ++depth) {
    for (size_t k = 0; k < cur.size(); ++k) {
      Ray& r = cur[k];
      float tmin = 1e38f;
      glm::vec3 bp(0.0f), bn(0.0f, 1.0f, 0.0f);
      bool hit = false;
      for (int gi = 0; gi < G; ++gi) {
        glm::vec3 p, n;
        bool outside = true;
        float t = geomTest(geoms[gi], r, p, n, outside);
        if (t > 0.0f && t < tmin) tmin = t, bp = p, bn = n, hit = true;
      }
      Ray nr;
      if (hit) {
        nr.origin = bp + bn * 0.001f;
        if ((k + depth) & 1) {
          nr.direction = r.direction - 2.0f * glm::dot(r.direction, bn) * bn;
        } else {
          glm::vec3 v = bn + 0.999f * rng.unit();
          nr.direction = glm::normalize(v);
        }
      } else {  // escaped: restart from a random point looking into the scene
        nr.origin = glm::vec3(4.0f * rng.sym(), 5.0f + 4.0f * rng.sym(), 4.0f * rng.sym());
        nr.direction = rng.unit();
      }
      if (!(nr.direction.x == nr.direction.x)) nr.direction = rng.unit();  // NaN normal: keep the set finite
      r = nr;
      rays.push_back(nr);
    }
  }
  return rays;
}

static void putVec(std::vector<uint32_t>& w, const glm::vec3& v) {
  w.push_back(fbits(v.x));
  w.push_back(fbits(v.y));
  w.push_back(fbits(v.z));
}
static void putMat(std::vector<uint32_t>& w, const glm::mat4& m) {  // GLM memory order: column-major
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) w.push_back(fbits(m[c][r]));
}

static int modeIsect(const char* out, int nscenes, char** paths) {
  gold::File f;
  // utilhash vectors: structured values of the seed expression and random words
  {
    gold::Section& in = f.add("hash_in", 1);
    gold::Section& ou = f.add("hash_out", 1);
    Rng rng(0x9e3779b9u);
    std::vector<uint32_t> v = {0u, 1u, 12345u, 0x7fffffffu, 0x80000000u, 0xffffffffu};
    for (uint32_t d = 0; d < 16; ++d)
      for (uint32_t i : {1u, 2u, 16u, 999u, 5000u}) v.push_back((1u << 31) | (d << 22) | i);
    for (uint32_t p : {63u, 64u, 65535u, 65536u, 639999u, 2073599u}) v.push_back(p);
    while (v.size() < 2048) v.push_back(rng.next());
    for (uint32_t a : v) in.w.push_back(a), ou.w.push_back(utilhash(a));
  }
  // seed list: (iter, index, depth) -> h as in makeSeededRandomEngine (pathtrace.cu:205), utilhash = the reference's
  {
    gold::Section& in = f.add("seed_in", 3);
    gold::Section& ou = f.add("seed_out", 1);
    Rng rng(0x51ed270bu);
    const int iters[] = {1, 2, 3, 16, 17, 1000, 4999, 5000};
    std::vector<int> idx = {0, 1, 2, 63, 64, 65535, 65536, 639999, 2073599};
    while (idx.size() < 16) idx.push_back((int)(rng.next() % 2073600u));
    for (int iter : iters)
      for (int depth = 0; depth < 12; ++depth)
        for (int index : idx) {
          int h = utilhash((1 << 31) | (depth << 22) | iter) ^ utilhash(index);
          in.w.push_back((uint32_t)iter), in.w.push_back((uint32_t)index), in.w.push_back((uint32_t)depth);
          ou.w.push_back((uint32_t)h);
        }
  }
  gold::Section& sets = f.add("sets", 2);  // per scene: number of geoms, number of rays
  for (int s = 0; s < nscenes; ++s) {
    Scene* scene = new Scene(paths[s]);  // never deleted: scene.h:21 declares a destructor scene.cpp never defines
    const std::vector<Geom>& geoms = scene->geoms;
    std::vector<Ray> rays = makeRays(geoms, 1000u + 77u * (uint32_t)s);
    sets.w.push_back((uint32_t)geoms.size());
    sets.w.push_back((uint32_t)rays.size());
    char name[24];
    snprintf(name, sizeof name, "geoms_%d", s);  // type, materialid, transform, inverseTransform, invTranspose
    gold::Section& gs = f.add(name, 2 + 48);
    for (const Geom& g : geoms) {
      gs.w.push_back((uint32_t)g.type);
      gs.w.push_back((uint32_t)g.materialid);
      putMat(gs.w, g.transform), putMat(gs.w, g.inverseTransform), putMat(gs.w, g.invTranspose);
    }
    snprintf(name, sizeof name, "rays_%d", s);  // origin, direction
    gold::Section& rs = f.add(name, 6);
    for (const Ray& r : rays) putVec(rs.w, r.origin), putVec(rs.w, r.direction);
    snprintf(name, sizeof name, "hits_%d", s);  // row = ray * G + geom: t, point, normal, outside
    gold::Section& hs = f.add(name, 8);
    for (const Ray& r : rays)
      for (const Geom& g : geoms) {
        glm::vec3 p(0.0f), n(0.0f);
        bool outside = false;
        float t = geomTest(g, r, p, n, outside);
        if (t == -1.0f) p = glm::vec3(0.0f), n = glm::vec3(0.0f), outside = false;  // outputs undefined on a miss
        hs.w.push_back(fbits(t));
        putVec(hs.w, p), putVec(hs.w, n);
        hs.w.push_back(outside ? 1u : 0u);
      }
  }
  return f.write(out) ? 0 : 1;
}

static void jvec(FILE* o, const char* name, const float* v, int n, const char* tail) {
  fprintf(o, "\"%s\": [", name);
  for (int i = 0; i < n; ++i) fprintf(o, "%u%s", fbits(v[i]), i + 1 < n ? ", " : "");
  fprintf(o, "]%s", tail);
}

static int modeScene(const char* out, int nscenes, char** paths) {
  FILE* o = fopen(out, "w");
  if (!o) return 1;
  fprintf(o, "{\n \"source\": \"reference src/scene.cpp + src/utilities.cpp compiled in place (new Scene(path), main.cpp:45); "
             "floats as bit patterns, matrices in GLM memory order (column-major)\",\n \"scenes\": [\n");
  for (int s = 0; s < nscenes; ++s) {
    Scene* scene = new Scene(paths[s]);
    const char* base = strrchr(paths[s], '/');
    fprintf(o, "  {\"file\": \"%s\",\n   \"geoms\": [\n", base ? base + 1 : paths[s]);
    for (size_t i = 0; i < scene->geoms.size(); ++i) {
      const Geom& g = scene->geoms[i];
      fprintf(o, "    {\"type\": %d, \"materialid\": %d, ", (int)g.type, g.materialid);
      jvec(o, "translation", &g.translation.x, 3, ", ");
      jvec(o, "rotation", &g.rotation.x, 3, ", ");
      jvec(o, "scale", &g.scale.x, 3, ",\n     ");
      jvec(o, "transform", &g.transform[0][0], 16, ",\n     ");
      jvec(o, "inverseTransform", &g.inverseTransform[0][0], 16, ",\n     ");
      jvec(o, "invTranspose", &g.invTranspose[0][0], 16, "}");
      fprintf(o, "%s\n", i + 1 < scene->geoms.size() ? "," : "");
    }
    fprintf(o, "   ],\n   \"materials\": [\n");
    for (size_t i = 0; i < scene->materials.size(); ++i) {
      const Material& m = scene->materials[i];
      fprintf(o, "    {");
      jvec(o, "color", &m.color.x, 3, ", ");
      jvec(o, "specular_exponent", &m.specular.exponent, 1, ", ");
      jvec(o, "specular_color", &m.specular.color.x, 3, ", ");
      jvec(o, "hasReflective", &m.hasReflective, 1, ", ");
      jvec(o, "hasRefractive", &m.hasRefractive, 1, ", ");
      jvec(o, "indexOfRefraction", &m.indexOfRefraction, 1, ", ");
      jvec(o, "emittance", &m.emittance, 1, "}");
      fprintf(o, "%s\n", i + 1 < scene->materials.size() ? "," : "");
    }
    const Camera& c = scene->state.camera;
    fprintf(o, "   ],\n   \"camera\": {\"resolution\": [%d, %d], ", c.resolution.x, c.resolution.y);
    jvec(o, "position", &c.position.x, 3, ", ");
    jvec(o, "lookAt", &c.lookAt.x, 3, ", ");
    jvec(o, "view", &c.view.x, 3, ",\n     ");
    jvec(o, "up", &c.up.x, 3, ", ");
    jvec(o, "right", &c.right.x, 3, ", ");
    jvec(o, "fov", &c.fov.x, 2, ", ");
    jvec(o, "pixelLength", &c.pixelLength.x, 2, "},\n");
    fprintf(o, "   \"iterations\": %u, \"traceDepth\": %d, \"imageName\": \"%s\", \"image_size\": %zu}%s\n",
            scene->state.iterations, scene->state.traceDepth, scene->state.imageName.c_str(), scene->state.image.size(),
            s + 1 < nscenes ? "," : "");
  }
  fprintf(o, " ]\n}\n");
  return fclose(o) == 0 ? 0 : 1;
}

int main(int argc, char** argv) {
  if (argc < 4) {
    fprintf(stderr, "usage: ref_hot scene|isect OUT SCENE.txt...\n");
    return 2;
  }
  if (!strcmp(argv[1], "scene")) return modeScene(argv[2], argc - 3, argv + 3);
  if (!strcmp(argv[1], "isect")) return modeIsect(argv[2], argc - 3, argv + 3);
  return 2;
}
