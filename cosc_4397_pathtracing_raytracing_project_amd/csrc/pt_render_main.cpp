// pt_render — headless driver: what the reference's main() / runCuda() / saveImage()
// (src/main.cpp:34-156) do without the GLFW window: load a scene file, run
// state.iterations iterations through pathtraceInit/pathtrace/pathtraceFree, write
// <FILE>.<spp>samp.png.  Flags exist only because the reference takes resolution,
// iteration count and depth from the scene file (scene.cpp:103-114).
//
//   pt_render SCENE.txt [--res WxH] [--spp N] [--depth D] [--out PREFIX] [--pfm]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/pathtrace_amd.hpp"
#include "pt_scene.h"

int main(int argc, char** argv) {
  if (argc < 2) {
    std::printf("Usage: %s SCENEFILE.txt [--res WxH] [--spp N] [--depth D] [--out PREFIX] [--pfm]\n", argv[0]);
    return 1;
  }
  int rw = 0, rh = 0, spp = 0, depth = 0;
  bool pfm = false;
  std::string out;
  for (int i = 2; i < argc; ++i) {
    if (!std::strcmp(argv[i], "--res") && i + 1 < argc) std::sscanf(argv[++i], "%dx%d", &rw, &rh);
    else if (!std::strcmp(argv[i], "--spp") && i + 1 < argc) spp = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "--depth") && i + 1 < argc) depth = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "--out") && i + 1 < argc) out = argv[++i];
    else if (!std::strcmp(argv[i], "--pfm")) pfm = true;
    else {
      std::fprintf(stderr, "unknown argument %s\n", argv[i]);
      return 1;
    }
  }
  pt::Scene* scene = nullptr;
  try {
    scene = new pt::Scene(argv[1]);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "%s\n", e.what());
    return 1;
  }
  if (rw > 0 && rh > 0) scene->overrideResolution(rw, rh);
  if (spp > 0) scene->state.iterations = spp;
  if (depth > 0) scene->state.traceDepth = depth;
  scene->applyInitialCameraState();
  const int W = scene->state.camera.resolution[0], H = scene->state.camera.resolution[1];
  const int iters = (int)scene->state.iterations;

  GuiDataContainer gui;
  InitDataContainer(&gui);
  pathtraceFree();  // main.cpp:134 frees before the first init
  pathtraceInit(scene);
  const auto t0 = std::chrono::high_resolution_clock::now();
  for (int it = 1; it <= iters; ++it) pathtrace(nullptr, 0, it);  // main.cpp:138-149
  pathtraceSyncImage();
  const double secs = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
  std::printf("%dx%d, %d spp, depth %d: %.3f s, %.2f Msamples/s\n", W, H, iters, scene->state.traceDepth, secs,
              (double)W * H * iters / secs / 1e6);
  if (out.empty()) out = scene->state.imageName;
  const std::string base = out + "." + std::to_string(iters) + "samp";
  if (pt_save_png((base + ".png").c_str(), scene->state.image.data(), W, H, (float)iters) == 0)
    std::printf("Saved %s.png.\n", base.c_str());
  if (pfm && pt_save_pfm((base + ".pfm").c_str(), scene->state.image.data(), W, H, (float)iters) == 0)
    std::printf("Saved %s.pfm.\n", base.c_str());
  pathtraceFree();
  delete scene;
  return 0;
}
