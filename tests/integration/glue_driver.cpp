// glue_driver.cpp — headless stand-in for the reference's main.cpp / preview.cpp around the glue (TEST
// INFRASTRUCTURE).  It performs what main() and runCuda() do between `new Scene` and saveImage, minus the window:
//   main.cpp:45        scene = new Scene(file)                       (the reference's own loader, linked in)
//   main.cpp:57-71     orbit state phi / theta / zoom from the loaded camera
//   main.cpp:110-128   the camera fix-up of the first frame (camchanged)
//   main.cpp:133-149   pathtraceFree(); pathtraceInit(scene); then pathtrace(pbo = NULL, 0, iteration) per iteration
//   main.cpp:152       pathtraceFree()
// and writes scene->state.image (the running SUM the renderer leaves there, pathtrace.cu:648-651) as raw floats.
// The camera arithmetic below is written against main.cpp with the same headers in scope (scene.h's `using namespace
// std`, GLM), so the overloads resolve as they do there; the GPU test compares the image bit for bit with the one the
// product's own loader + fix-up (pt_scene_load) renders, which pins this restatement as well.
//
//   ref_glue_demo SCENE.txt OUT.f32 [ITERATIONS] [W H]
#include <cstdio>
#include <cstdlib>

#include <glm/glm.hpp>

#include "scene.h"
#include "pathtrace.h"

int main(int argc, char** argv) {
  if (argc < 3) {
    fprintf(stderr, "usage: %s SCENE.txt OUT.f32 [ITERATIONS] [W H]\n", argv[0]);
    return 2;
  }
  Scene* scene = new Scene(argv[1]);
  GuiDataContainer* guiData = new GuiDataContainer();
  RenderState* renderState = &scene->state;
  Camera& cam = renderState->camera;
  if (argc > 3) renderState->iterations = (unsigned)atoi(argv[3]);
  if (argc > 5) {  // resolution override: recompute what scene.cpp:133-140 derives from RES
    float fovy = cam.fov.y;
    cam.resolution.x = atoi(argv[4]);
    cam.resolution.y = atoi(argv[5]);
    float yscaled = tan(fovy * (PI / 180));
    float xscaled = (yscaled * cam.resolution.x) / cam.resolution.y;
    float fovx = (atan(xscaled) * 180) / PI;
    cam.fov = glm::vec2(fovx, fovy);
    cam.pixelLength = glm::vec2(2 * xscaled / (float)cam.resolution.x, 2 * yscaled / (float)cam.resolution.y);
    renderState->image.assign((size_t)cam.resolution.x * cam.resolution.y, glm::vec3());
  }
  // main.cpp:57-71
  glm::vec3 view = cam.view;
  glm::vec3 viewXZ = glm::vec3(view.x, 0.0f, view.z);
  glm::vec3 viewZY = glm::vec3(0.0f, view.y, view.z);
  float phi = glm::acos(glm::dot(glm::normalize(viewXZ), glm::vec3(0, 0, -1)));
  float theta = glm::acos(glm::dot(glm::normalize(viewZY), glm::vec3(0, 1, 0)));
  float zoom = glm::length(cam.position - cam.lookAt);
  // main.cpp:110-128
  glm::vec3 cameraPosition;
  cameraPosition.x = zoom * sin(phi) * sin(theta);
  cameraPosition.y = zoom * cos(theta);
  cameraPosition.z = zoom * cos(phi) * sin(theta);
  cam.view = -glm::normalize(cameraPosition);
  glm::vec3 v = cam.view;
  glm::vec3 u = glm::vec3(0, 1, 0);
  glm::vec3 r = glm::cross(v, u);
  cam.up = glm::cross(r, v);
  cam.right = r;
  cameraPosition += cam.lookAt;
  cam.position = cameraPosition;

  InitDataContainer(guiData);
  pathtraceFree();  // main.cpp:134: Free comes BEFORE the first Init
  pathtraceInit(scene);
  for (unsigned iteration = 1; iteration <= renderState->iterations; ++iteration) pathtrace(NULL, 0, (int)iteration);
  FILE* f = fopen(argv[2], "wb");
  if (!f) return 1;
  fwrite(renderState->image.data(), sizeof(glm::vec3), renderState->image.size(), f);
  fclose(f);
  pathtraceFree();
  pathtraceFree();  // twice in a row must be harmless
  fprintf(stderr, "glue demo: %d x %d, %u iterations, traced depth %d\n", cam.resolution.x, cam.resolution.y,
          renderState->iterations, guiData->TracedDepth);
  return 0;
}
