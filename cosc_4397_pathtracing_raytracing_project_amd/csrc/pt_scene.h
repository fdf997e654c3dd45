// pt_scene.h — host-side scene model of the MI355X path tracer (product code).
//
// Mirrors the public surface of the reference's `Scene` (src/scene.h:20-25:
// geoms, materials, state) with plain structs from include/pt_amd.h so it can cross
// the C ABI, and restates the arithmetic of the reference's loader in GLM 0.9.6
// operation order so the transform/inverse/invTranspose matrices and the camera
// basis are bit-identical to what `new Scene(file)` + main.cpp produce.
#pragma once
#include <string>
#include <vector>

#include "../../include/pt_amd.h"

namespace pt {

struct RenderState {  // src/sceneStructs.h:61-67
  PtCamera camera{};
  unsigned int iterations = 0;
  int traceDepth = 0;
  std::vector<float> image;  // W*H*3 running sum
  std::string imageName;
};

class Scene {  // src/scene.h:11-26
 public:
  explicit Scene(const std::string& filename);  // throws std::runtime_error if unreadable
  std::vector<PtGeom> geoms;
  std::vector<PtMaterial> materials;
  RenderState state;
  float fovy = 0.0f;

  // scene.cpp:133-140 for a new resolution (headless stand-in for editing RES).
  void overrideResolution(int w, int h);
  // main.cpp:57-71 + 110-128: what the first runCuda() does to the camera.
  void applyInitialCameraState();
  PtSceneDesc desc() const;
};

// utilities.cpp:64-72 / scene.cpp:83-86 restated (exposed for tests).
void buildTransform(const float trs[9], float transform[16], float inverse[16], float invTranspose[16]);

// pathtrace.cu:34-111.
void buildBVH(const PtGeom* geoms, int n, std::vector<PtBVHNode>& nodes);

}  // namespace pt
