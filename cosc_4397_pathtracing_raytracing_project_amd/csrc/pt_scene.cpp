// pt_scene.cpp — scene text parser, transform builder, camera state and BVH
// builder of the MI355X path tracer (host C++; product code).
//
// Behavioural source: the reference's src/scene.cpp, src/utilities.cpp,
// src/main.cpp:57-71,110-128 and src/pathtrace.cu:34-111.  Nothing here is shared
// with oracle/; tests compare the two implementations table-by-table.
// All float arithmetic follows GLM 0.9.6.3's operation order (the reference's
// vendored copy) because the matrices feed a bit-exact parity test.
// Must be compiled with -ffp-contract=off.
#include "pt_scene.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <sstream>
#include <stdexcept>

namespace pt {
namespace {

// ---- tiny column-major 4x4 helpers; E(m,c,r) == glm m[c][r] ------------------
struct M4 {
  float e[16];
};
inline float& E(M4& m, int c, int r) { return m.e[c * 4 + r]; }
inline float E(const M4& m, int c, int r) { return m.e[c * 4 + r]; }
M4 identity() {
  M4 m{};
  for (int i = 0; i < 4; ++i) E(m, i, i) = 1.0f;
  return m;
}
struct V3 {
  float x, y, z;
};
inline float dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }  // (x+y)+z
inline V3 cross3(V3 a, V3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
inline V3 scale3(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 normalize3(V3 a) { return scale3(a, 1.0f / std::sqrt(dot3(a, a))); }
inline float length3(V3 a) { return std::sqrt(dot3(a, a)); }

// column c of m scaled by s
inline void colScale(const M4& m, int c, float s, float out[4]) {
  for (int r = 0; r < 4; ++r) out[r] = E(m, c, r) * s;
}
// glm translate (gtc/matrix_transform.inl:40-49): col3 = m0*v0 + m1*v1 + m2*v2 + m3
M4 glmTranslate(const M4& m, V3 v) {
  M4 r = m;
  for (int k = 0; k < 4; ++k) E(r, 3, k) = ((E(m, 0, k) * v.x + E(m, 1, k) * v.y) + E(m, 2, k) * v.z) + E(m, 3, k);
  return r;
}
// glm rotate (gtc/matrix_transform.inl:52-85)
M4 glmRotate(const M4& m, float angle, V3 axisIn) {
  const float c = std::cos(angle), s = std::sin(angle);  // float overloads
  V3 ax = normalize3(axisIn);
  float axis[3] = {ax.x, ax.y, ax.z};
  float temp[3] = {(1.0f - c) * ax.x, (1.0f - c) * ax.y, (1.0f - c) * ax.z};
  float R[3][3];
  R[0][0] = c + temp[0] * axis[0];
  R[0][1] = 0 + temp[0] * axis[1] + s * axis[2];
  R[0][2] = 0 + temp[0] * axis[2] - s * axis[1];
  R[1][0] = 0 + temp[1] * axis[0] - s * axis[2];
  R[1][1] = c + temp[1] * axis[1];
  R[1][2] = 0 + temp[1] * axis[2] + s * axis[0];
  R[2][0] = 0 + temp[2] * axis[0] + s * axis[1];
  R[2][1] = 0 + temp[2] * axis[1] - s * axis[0];
  R[2][2] = c + temp[2] * axis[2];
  M4 out{};
  for (int j = 0; j < 3; ++j)
    for (int k = 0; k < 4; ++k) E(out, j, k) = (E(m, 0, k) * R[j][0] + E(m, 1, k) * R[j][1]) + E(m, 2, k) * R[j][2];
  for (int k = 0; k < 4; ++k) E(out, 3, k) = E(m, 3, k);
  return out;
}
// glm scale (gtc/matrix_transform.inl:122-134)
M4 glmScale(const M4& m, V3 v) {
  M4 out{};
  const float s[3] = {v.x, v.y, v.z};
  for (int j = 0; j < 3; ++j)
    for (int k = 0; k < 4; ++k) E(out, j, k) = E(m, j, k) * s[j];
  for (int k = 0; k < 4; ++k) E(out, 3, k) = E(m, 3, k);
  return out;
}
// glm mat4*mat4 (detail/type_mat4x4.inl:686-704)
M4 glmMul(const M4& a, const M4& b) {
  M4 out{};
  for (int j = 0; j < 4; ++j)
    for (int k = 0; k < 4; ++k)
      E(out, j, k) = ((E(a, 0, k) * E(b, j, 0) + E(a, 1, k) * E(b, j, 1)) + E(a, 2, k) * E(b, j, 2)) + E(a, 3, k) * E(b, j, 3);
  return out;
}
// glm inverse (detail/type_mat4x4.inl:37-92)
M4 glmInverse(const M4& m) {
#define MM(c, r) E(m, c, r)
  const float c00 = MM(2, 2) * MM(3, 3) - MM(3, 2) * MM(2, 3), c02 = MM(1, 2) * MM(3, 3) - MM(3, 2) * MM(1, 3),
              c03 = MM(1, 2) * MM(2, 3) - MM(2, 2) * MM(1, 3), c04 = MM(2, 1) * MM(3, 3) - MM(3, 1) * MM(2, 3),
              c06 = MM(1, 1) * MM(3, 3) - MM(3, 1) * MM(1, 3), c07 = MM(1, 1) * MM(2, 3) - MM(2, 1) * MM(1, 3),
              c08 = MM(2, 1) * MM(3, 2) - MM(3, 1) * MM(2, 2), c10 = MM(1, 1) * MM(3, 2) - MM(3, 1) * MM(1, 2),
              c11 = MM(1, 1) * MM(2, 2) - MM(2, 1) * MM(1, 2), c12 = MM(2, 0) * MM(3, 3) - MM(3, 0) * MM(2, 3),
              c14 = MM(1, 0) * MM(3, 3) - MM(3, 0) * MM(1, 3), c15 = MM(1, 0) * MM(2, 3) - MM(2, 0) * MM(1, 3),
              c16 = MM(2, 0) * MM(3, 2) - MM(3, 0) * MM(2, 2), c18 = MM(1, 0) * MM(3, 2) - MM(3, 0) * MM(1, 2),
              c19 = MM(1, 0) * MM(2, 2) - MM(2, 0) * MM(1, 2), c20 = MM(2, 0) * MM(3, 1) - MM(3, 0) * MM(2, 1),
              c22 = MM(1, 0) * MM(3, 1) - MM(3, 0) * MM(1, 1), c23 = MM(1, 0) * MM(2, 1) - MM(2, 0) * MM(1, 1);
  const float F0[4] = {c00, c00, c02, c03}, F1[4] = {c04, c04, c06, c07}, F2[4] = {c08, c08, c10, c11},
              F3[4] = {c12, c12, c14, c15}, F4[4] = {c16, c16, c18, c19}, F5[4] = {c20, c20, c22, c23};
  const float V0[4] = {MM(1, 0), MM(0, 0), MM(0, 0), MM(0, 0)}, V1[4] = {MM(1, 1), MM(0, 1), MM(0, 1), MM(0, 1)},
              V2[4] = {MM(1, 2), MM(0, 2), MM(0, 2), MM(0, 2)}, V3_[4] = {MM(1, 3), MM(0, 3), MM(0, 3), MM(0, 3)};
  const float SA[4] = {+1, -1, +1, -1}, SB[4] = {-1, +1, -1, +1};
  M4 inv{};
  for (int k = 0; k < 4; ++k) {
    E(inv, 0, k) = ((V1[k] * F0[k] - V2[k] * F1[k]) + V3_[k] * F2[k]) * SA[k];
    E(inv, 1, k) = ((V0[k] * F0[k] - V2[k] * F3[k]) + V3_[k] * F4[k]) * SB[k];
    E(inv, 2, k) = ((V0[k] * F1[k] - V1[k] * F3[k]) + V3_[k] * F5[k]) * SA[k];
    E(inv, 3, k) = ((V0[k] * F2[k] - V1[k] * F4[k]) + V2[k] * F5[k]) * SB[k];
  }
  const float d0 = MM(0, 0) * E(inv, 0, 0), d1 = MM(0, 1) * E(inv, 1, 0), d2 = MM(0, 2) * E(inv, 2, 0),
              d3 = MM(0, 3) * E(inv, 3, 0);
  const float oneOverDet = 1.0f / ((d0 + d1) + (d2 + d3));
  for (float& v : inv.e) v = v * oneOverDet;
  return inv;
#undef MM
}
// glm inverseTranspose (gtc/matrix_inverse.inl:95-147)
M4 glmInverseTranspose(const M4& m) {
#define MM(c, r) E(m, c, r)
  const float S00 = MM(2, 2) * MM(3, 3) - MM(3, 2) * MM(2, 3), S01 = MM(2, 1) * MM(3, 3) - MM(3, 1) * MM(2, 3),
              S02 = MM(2, 1) * MM(3, 2) - MM(3, 1) * MM(2, 2), S03 = MM(2, 0) * MM(3, 3) - MM(3, 0) * MM(2, 3),
              S04 = MM(2, 0) * MM(3, 2) - MM(3, 0) * MM(2, 2), S05 = MM(2, 0) * MM(3, 1) - MM(3, 0) * MM(2, 1),
              S06 = MM(1, 2) * MM(3, 3) - MM(3, 2) * MM(1, 3), S07 = MM(1, 1) * MM(3, 3) - MM(3, 1) * MM(1, 3),
              S08 = MM(1, 1) * MM(3, 2) - MM(3, 1) * MM(1, 2), S09 = MM(1, 0) * MM(3, 3) - MM(3, 0) * MM(1, 3),
              S10 = MM(1, 0) * MM(3, 2) - MM(3, 0) * MM(1, 2), S11 = MM(1, 1) * MM(3, 3) - MM(3, 1) * MM(1, 3),
              S12 = MM(1, 0) * MM(3, 1) - MM(3, 0) * MM(1, 1), S13 = MM(1, 2) * MM(2, 3) - MM(2, 2) * MM(1, 3),
              S14 = MM(1, 1) * MM(2, 3) - MM(2, 1) * MM(1, 3), S15 = MM(1, 1) * MM(2, 2) - MM(2, 1) * MM(1, 2),
              S16 = MM(1, 0) * MM(2, 3) - MM(2, 0) * MM(1, 3), S17 = MM(1, 0) * MM(2, 2) - MM(2, 0) * MM(1, 2),
              S18 = MM(1, 0) * MM(2, 1) - MM(2, 0) * MM(1, 1);
  M4 I{};
  E(I, 0, 0) = +((MM(1, 1) * S00 - MM(1, 2) * S01) + MM(1, 3) * S02);
  E(I, 0, 1) = -((MM(1, 0) * S00 - MM(1, 2) * S03) + MM(1, 3) * S04);
  E(I, 0, 2) = +((MM(1, 0) * S01 - MM(1, 1) * S03) + MM(1, 3) * S05);
  E(I, 0, 3) = -((MM(1, 0) * S02 - MM(1, 1) * S04) + MM(1, 2) * S05);
  E(I, 1, 0) = -((MM(0, 1) * S00 - MM(0, 2) * S01) + MM(0, 3) * S02);
  E(I, 1, 1) = +((MM(0, 0) * S00 - MM(0, 2) * S03) + MM(0, 3) * S04);
  E(I, 1, 2) = -((MM(0, 0) * S01 - MM(0, 1) * S03) + MM(0, 3) * S05);
  E(I, 1, 3) = +((MM(0, 0) * S02 - MM(0, 1) * S04) + MM(0, 2) * S05);
  E(I, 2, 0) = +((MM(0, 1) * S06 - MM(0, 2) * S07) + MM(0, 3) * S08);
  E(I, 2, 1) = -((MM(0, 0) * S06 - MM(0, 2) * S09) + MM(0, 3) * S10);
  E(I, 2, 2) = +((MM(0, 0) * S11 - MM(0, 1) * S09) + MM(0, 3) * S12);
  E(I, 2, 3) = -((MM(0, 0) * S08 - MM(0, 1) * S10) + MM(0, 2) * S12);
  E(I, 3, 0) = -((MM(0, 1) * S13 - MM(0, 2) * S14) + MM(0, 3) * S15);
  E(I, 3, 1) = +((MM(0, 0) * S13 - MM(0, 2) * S16) + MM(0, 3) * S17);
  E(I, 3, 2) = -((MM(0, 0) * S14 - MM(0, 1) * S16) + MM(0, 3) * S18);
  E(I, 3, 3) = +((MM(0, 0) * S15 - MM(0, 1) * S17) + MM(0, 2) * S18);
  const float det = ((+MM(0, 0) * E(I, 0, 0) + MM(0, 1) * E(I, 0, 1)) + MM(0, 2) * E(I, 0, 2)) + MM(0, 3) * E(I, 0, 3);
  for (float& v : I.e) v = v / det;
  return I;
#undef MM
}

const float kPi = 3.1415926535897932384626422832795028841971f;  // utilities.h:12 (float literal)

// ---- text helpers (utilities.cpp:70-112) -----------------------------------
// Line reader accepting \n, \r\n and \r; a final line without terminator is still
// returned; eofbit is raised only by an empty read at EOF.
std::istream& readLine(std::istream& is, std::string& out) {
  out.clear();
  std::istream::sentry guard(is, true);
  std::streambuf* buf = is.rdbuf();
  while (true) {
    const int ch = buf->sbumpc();
    if (ch == '\n') break;
    if (ch == '\r') {
      if (buf->sgetc() == '\n') buf->sbumpc();
      break;
    }
    if (ch == std::streambuf::traits_type::eof()) {
      if (out.empty()) is.setstate(std::ios::eofbit);
      break;
    }
    out.push_back(static_cast<char>(ch));
  }
  return is;
}
std::vector<std::string> splitWords(const std::string& s) {
  std::vector<std::string> words;
  std::istringstream ss(s);
  for (std::string w; ss >> w;) words.push_back(w);
  return words;
}
float numAt(const std::vector<std::string>& w, size_t i) { return i < w.size() ? (float)atof(w[i].c_str()) : 0.0f; }
void read3(const std::vector<std::string>& w, float out[3]) {
  out[0] = numAt(w, 1), out[1] = numAt(w, 2), out[2] = numAt(w, 3);
}

// MATERIAL block: exactly 7 property lines (scene.cpp:153-188)
void parseMaterial(std::istream& in, const std::string& idWord, std::vector<PtMaterial>& mats) {
  if (atoi(idWord.c_str()) != (int)mats.size()) return;  // out-of-sequence id: block ignored (scene.cpp:155-157)
  PtMaterial m{};
  std::string line;
  for (int i = 0; i < 7; ++i) {
    readLine(in, line);
    auto w = splitWords(line);
    if (w.empty()) continue;
    const std::string& k = w[0];
    if (k == "RGB") read3(w, m.color);
    else if (k == "SPECEX") m.specular_exponent = numAt(w, 1);
    else if (k == "SPECRGB") read3(w, m.specular_color);
    else if (k == "REFL") m.hasReflective = numAt(w, 1);
    else if (k == "REFR") m.hasRefractive = numAt(w, 1);
    else if (k == "REFRIOR") m.indexOfRefraction = numAt(w, 1);
    else if (k == "EMITTANCE") m.emittance = numAt(w, 1);
  }
  mats.push_back(m);
}
// OBJECT block: type line, "material N", then TRANS/ROTAT/SCALE until a blank line (scene.cpp:35-90)
//
// Mesh EXTENSION (the format names "mesh", INSTRUCTION.md:246, the reference implements neither loader
// nor kernel for it): an OBJECT of type `mesh` may carry `TRI x0 y0 z0 x1 y1 z1 x2 y2 z2` lines (object space) among its
// TRANS / ROTAT / SCALE lines — lines the reference's loader skips (scene.cpp:66-80) — and expands into one PT_GEOM_TRIANGLE
// primitive per line, vertices transformed to world space with the object's matrix (no TRI line: one ordinary geom, see
// below).  OBJECT ids are checked against the
// number of objects accepted so far (`objects`), which equals geoms.size() (the reference's check, scene.cpp:37) as long as
// no mesh has expanded.
void parseObject(std::istream& in, const std::string& idWord, std::vector<PtGeom>& geoms, int& objects) {
  if (atoi(idWord.c_str()) != objects) return;
  ++objects;
  PtGeom g{};
  g.type = PT_GEOM_SPHERE;
  float trs[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // GLM vec3 members default to zero
  std::string line;
  bool mesh = false;
  std::vector<float> tri;  // 9 floats per TRI line
  readLine(in, line);
  if (!line.empty() && in.good()) {
    if (line == "sphere") g.type = PT_GEOM_SPHERE;
    else if (line == "cube") g.type = PT_GEOM_CUBE;
    else if (line == "mesh") mesh = true;
  }
  readLine(in, line);
  if (!line.empty() && in.good()) {
    auto w = splitWords(line);
    if (w.size() > 1) g.materialid = atoi(w[1].c_str());
  }
  for (readLine(in, line); !line.empty() && in.good(); readLine(in, line)) {
    auto w = splitWords(line);
    if (w.empty()) continue;
    if (w[0] == "TRANS") read3(w, trs + 0);
    else if (w[0] == "ROTAT") read3(w, trs + 3);
    else if (w[0] == "SCALE") read3(w, trs + 6);
    else if (mesh && w[0] == "TRI" && w.size() >= 10)
      for (size_t k = 1; k <= 9; ++k) tri.push_back(numAt(w, k));
  }
  buildTransform(trs, g.transform, g.inverseTransform, g.invTranspose);
  // The extension acts only on its own syntax: a `mesh` block WITHOUT TRI lines — all a reference-format file can
  // contain — loads as the reference's loader leaves it: ONE geom whose type keeps its initial value (scene.cpp:47-55
  // assigns none for an unknown type line; uninitialised there, PT_GEOM_SPHERE here), so the geom index space of such
  // a file does not shift.
  if (mesh && !tri.empty()) {
    for (size_t k = 0; k + 8 < tri.size(); k += 9) {
      PtGeom t{};
      t.type = PT_GEOM_TRIANGLE;
      t.materialid = g.materialid;
      for (int v = 0; v < 3; ++v) {
        const float* p = &tri[k + 3 * v];
        const float* m = g.transform;  // vec3(transform * vec4(p, 1)) in GLM order: (m0*x + m1*y) + (m2*z + m3*1)
        for (int r = 0; r < 3; ++r)
          t.transform[3 * v + r] = (m[0 * 4 + r] * p[0] + m[1 * 4 + r] * p[1]) + (m[2 * 4 + r] * p[2] + m[3 * 4 + r] * 1.0f);
      }
      geoms.push_back(t);
    }
    return;
  }
  geoms.push_back(g);
}

void cameraScale(PtCamera& cam, float fovy) {  // scene.cpp:133-140
  const float yscaled = std::tan(fovy * (kPi / 180));
  const float xscaled = (yscaled * cam.resolution[0]) / cam.resolution[1];
  const float fovx = (std::atan(xscaled) * 180) / kPi;
  cam.fov[0] = fovx;
  cam.fov[1] = fovy;
  cam.pixelLength[0] = 2 * xscaled / (float)cam.resolution[0];
  cam.pixelLength[1] = 2 * yscaled / (float)cam.resolution[1];
}
// CAMERA block (scene.cpp:92-151)
void parseCamera(std::istream& in, Scene& sc) {
  PtCamera& cam = sc.state.camera;
  std::string line;
  for (int i = 0; i < 5; ++i) {
    readLine(in, line);
    auto w = splitWords(line);
    if (w.empty()) continue;
    if (w[0] == "RES" && w.size() > 2) {
      cam.resolution[0] = atoi(w[1].c_str());
      cam.resolution[1] = atoi(w[2].c_str());
    } else if (w[0] == "FOVY") sc.fovy = numAt(w, 1);
    else if (w[0] == "ITERATIONS" && w.size() > 1) sc.state.iterations = atoi(w[1].c_str());
    else if (w[0] == "DEPTH" && w.size() > 1) sc.state.traceDepth = atoi(w[1].c_str());
    else if (w[0] == "FILE" && w.size() > 1) sc.state.imageName = w[1];
  }
  for (readLine(in, line); !line.empty() && in.good(); readLine(in, line)) {
    auto w = splitWords(line);
    if (w.empty()) continue;
    if (w[0] == "EYE") read3(w, cam.position);
    else if (w[0] == "LOOKAT") read3(w, cam.lookAt);
    else if (w[0] == "UP") read3(w, cam.up);
  }
  cameraScale(cam, sc.fovy);
  // The reference derives `right` from `view` before `view` is assigned (scene.cpp:138
  // vs :142), i.e. from the zero vector: NaN.  Kept, because applyInitialCameraState()
  // overwrites it exactly as main.cpp does.
  V3 view0{cam.view[0], cam.view[1], cam.view[2]}, up{cam.up[0], cam.up[1], cam.up[2]};
  V3 r = normalize3(cross3(view0, up));
  cam.right[0] = r.x, cam.right[1] = r.y, cam.right[2] = r.z;
  V3 v = normalize3({cam.lookAt[0] - cam.position[0], cam.lookAt[1] - cam.position[1], cam.lookAt[2] - cam.position[2]});
  cam.view[0] = v.x, cam.view[1] = v.y, cam.view[2] = v.z;
  sc.state.image.assign((size_t)cam.resolution[0] * cam.resolution[1] * 3, 0.0f);
}

}  // namespace

void buildTransform(const float trs[9], float transform[16], float inverse[16], float invTranspose[16]) {
  const M4 I = identity();
  const M4 T = glmTranslate(I, {trs[0], trs[1], trs[2]});
  M4 R = glmRotate(I, trs[3] * (float)kPi / 180, {1, 0, 0});
  R = glmMul(R, glmRotate(I, trs[4] * (float)kPi / 180, {0, 1, 0}));
  R = glmMul(R, glmRotate(I, trs[5] * (float)kPi / 180, {0, 0, 1}));
  const M4 S = glmScale(I, {trs[6], trs[7], trs[8]});
  const M4 M = glmMul(glmMul(T, R), S);
  std::memcpy(transform, M.e, sizeof(M.e));
  const M4 inv = glmInverse(M);
  std::memcpy(inverse, inv.e, sizeof(inv.e));
  const M4 it = glmInverseTranspose(M);
  std::memcpy(invTranspose, it.e, sizeof(it.e));
}

Scene::Scene(const std::string& filename) {
  std::ifstream in(filename.c_str());
  if (!in.is_open()) throw std::runtime_error("cannot read scene file: " + filename);  // scene.cpp:12-15 aborts
  std::string line;
  int objects = 0;  // OBJECT blocks accepted (see parseObject)
  while (in.good()) {  // scene.cpp:16-32
    readLine(in, line);
    if (line.empty()) continue;
    auto w = splitWords(line);
    if (w.empty()) continue;
    if (w[0] == "MATERIAL") parseMaterial(in, w.size() > 1 ? w[1] : "", materials);
    else if (w[0] == "OBJECT") parseObject(in, w.size() > 1 ? w[1] : "", geoms, objects);
    else if (w[0] == "CAMERA") parseCamera(in, *this);
  }
}

void Scene::overrideResolution(int w, int h) {
  state.camera.resolution[0] = w;
  state.camera.resolution[1] = h;
  cameraScale(state.camera, fovy);
  state.image.assign((size_t)w * h * 3, 0.0f);
}

void Scene::applyInitialCameraState() {
  PtCamera& cam = state.camera;
  const V3 view{cam.view[0], cam.view[1], cam.view[2]};
  const V3 look{cam.lookAt[0], cam.lookAt[1], cam.lookAt[2]};
  const V3 pos{cam.position[0], cam.position[1], cam.position[2]};
  // main.cpp:63-70 — orbit angles and zoom from the loaded camera
  const V3 viewXZ = normalize3({view.x, 0.0f, view.z});
  const V3 viewZY = normalize3({0.0f, view.y, view.z});
  const float phi = std::acos(dot3(viewXZ, {0, 0, -1}));
  const float theta = std::acos(dot3(viewZY, {0, 1, 0}));
  const float zoom = length3({pos.x - look.x, pos.y - look.y, pos.z - look.z});
  // main.cpp:110-128 — first runCuda() with camchanged == true
  V3 cp;
  cp.x = zoom * std::sin(phi) * std::sin(theta);
  cp.y = zoom * std::cos(theta);
  cp.z = zoom * std::cos(phi) * std::sin(theta);
  const V3 n = normalize3(cp);
  const V3 v{-n.x, -n.y, -n.z};
  const V3 r = cross3(v, {0, 1, 0});
  const V3 u = cross3(r, v);
  cam.view[0] = v.x, cam.view[1] = v.y, cam.view[2] = v.z;
  cam.up[0] = u.x, cam.up[1] = u.y, cam.up[2] = u.z;
  cam.right[0] = r.x, cam.right[1] = r.y, cam.right[2] = r.z;
  cam.position[0] = cp.x + look.x, cam.position[1] = cp.y + look.y, cam.position[2] = cp.z + look.z;
}

PtSceneDesc Scene::desc() const {
  PtSceneDesc d{};
  d.geoms = geoms.data();
  d.num_geoms = (int)geoms.size();
  d.materials = materials.data();
  d.num_materials = (int)materials.size();
  d.camera = state.camera;
  d.trace_depth = state.traceDepth;
  return d;
}

// ---- BVH (pathtrace.cu:34-111) ----------------------------------------------
namespace {
struct Box {
  float lo[3], hi[3];
};
Box worldBounds(const PtGeom& g) {  // 8 transformed unit-cube corners
  Box b;
  if (g.type == PT_GEOM_TRIANGLE) {
    // mesh extension: the three world-space vertices, padded — an axis-aligned triangle has a flat box, which the
    // strict slab test (tmax <= tmin rejects, pathtrace.cu:113-128) would never let a ray into
    for (int a = 0; a < 3; ++a) {
      const float v0 = g.transform[a], v1 = g.transform[3 + a], v2 = g.transform[6 + a];
      float lo = v0 < v1 ? v0 : v1, hi = v0 > v1 ? v0 : v1;
      lo = lo < v2 ? lo : v2, hi = hi > v2 ? hi : v2;
      const float pad = 1e-4f * std::max(1.0f, std::max(std::fabs(lo), std::fabs(hi)));
      b.lo[a] = lo - pad, b.hi[a] = hi + pad;
    }
    return b;
  }
  for (int a = 0; a < 3; ++a) b.lo[a] = std::numeric_limits<float>::max(), b.hi[a] = -std::numeric_limits<float>::max();
  for (int i = 0; i < 8; ++i) {
    const float c[4] = {(i & 1) ? 0.5f : -0.5f, (i & 2) ? 0.5f : -0.5f, (i & 4) ? 0.5f : -0.5f, 1.0f};
    for (int r = 0; r < 3; ++r) {
      const float* m = g.transform;
      const float w = (m[0 * 4 + r] * c[0] + m[1 * 4 + r] * c[1]) + (m[2 * 4 + r] * c[2] + m[3 * 4 + r] * c[3]);
      b.lo[r] = b.lo[r] < w ? b.lo[r] : w;  // glm::min(a,b) = a<b ? a : b
      b.hi[r] = b.hi[r] > w ? b.hi[r] : w;
    }
  }
  return b;
}
int buildNode(const std::vector<Box>& boxes, std::vector<int>& order, int first, int last, std::vector<PtBVHNode>& nodes) {
  const int self = (int)nodes.size();
  nodes.push_back(PtBVHNode{});
  if (last - first == 1) {
    const Box& b = boxes[order[first]];
    PtBVHNode& n = nodes[self];
    std::memcpy(n.bmin, b.lo, 12);
    std::memcpy(n.bmax, b.hi, 12);
    n.left = n.right = -1;
    n.geomIndex = order[first];
    return self;
  }
  float clo[3], chi[3];
  for (int a = 0; a < 3; ++a) clo[a] = std::numeric_limits<float>::max(), chi[a] = -std::numeric_limits<float>::max();
  for (int i = first; i < last; ++i) {
    const Box& b = boxes[order[i]];
    for (int a = 0; a < 3; ++a) {
      const float c = (b.lo[a] + b.hi[a]) * 0.5f;
      clo[a] = clo[a] < c ? clo[a] : c;
      chi[a] = chi[a] > c ? chi[a] : c;
    }
  }
  const float ex = chi[0] - clo[0], ey = chi[1] - clo[1], ez = chi[2] - clo[2];
  const int axis = (ex > ey && ex > ez) ? 0 : (ey > ez) ? 1 : 2;
  // std::sort on purpose: tie order of equal centroids is libstdc++'s, as in the reference.
  std::sort(order.begin() + first, order.begin() + last, [&](int a, int b) {
    const float ca = (boxes[a].lo[axis] + boxes[a].hi[axis]) * 0.5f;
    const float cb = (boxes[b].lo[axis] + boxes[b].hi[axis]) * 0.5f;
    return ca < cb;
  });
  const int mid = first + (last - first) / 2;
  const int l = buildNode(boxes, order, first, mid, nodes);
  const int r = buildNode(boxes, order, mid, last, nodes);
  PtBVHNode& n = nodes[self];
  n.left = l, n.right = r, n.geomIndex = -1;
  for (int a = 0; a < 3; ++a) {
    n.bmin[a] = nodes[l].bmin[a] < nodes[r].bmin[a] ? nodes[l].bmin[a] : nodes[r].bmin[a];
    n.bmax[a] = nodes[l].bmax[a] > nodes[r].bmax[a] ? nodes[l].bmax[a] : nodes[r].bmax[a];
  }
  return self;
}
}  // namespace

void buildBVH(const PtGeom* geoms, int n, std::vector<PtBVHNode>& nodes) {
  nodes.clear();
  if (n <= 0) return;
  std::vector<Box> boxes(n);
  for (int i = 0; i < n; ++i) boxes[i] = worldBounds(geoms[i]);
  std::vector<int> order(n);
  for (int i = 0; i < n; ++i) order[i] = i;
  nodes.reserve(2 * (size_t)n);
  buildNode(boxes, order, 0, n, nodes);
}

}  // namespace pt
