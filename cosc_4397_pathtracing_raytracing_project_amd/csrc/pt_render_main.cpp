// pt_render — headless driver: what the reference's main() / runCuda() / saveImage()
// (src/main.cpp:34-156) do without the GLFW window: load a scene file, run
// state.iterations iterations, write the PNG saveImage() would write.  Flags exist only because
// the reference takes resolution, iteration count and depth from the scene file (scene.cpp:103-114).
//
//   pt_render SCENE.txt [--res WxH] [--spp N] [--depth D] [--out PREFIX] [--pfm] [--hdr]
//                       [--arith exact|fma|fast] [--gpus K | --devices LIST] [--transport rccl|copy] [--stamp] [--aa]
//                       [--preview N]
//
// Without --gpus the run goes through the pathtrace.h-compatible shim (pathtraceInit / pathtrace per
// iteration / pathtraceFree), i.e. the code path a reference main.cpp would take.  With --gpus K (K >= 1;
// K = 0: all visible devices) it goes through pt_group_*: K devices in this one process, row-interleaved
// tiles, one grouped RCCL send/recv at write-out (BASELINE config 4), PNG bytes converted on the devices.
// --devices 0,0,0 names the devices explicitly; a device may appear more than once (several contexts on one GPU — the
// exchange then uses peer / device copies instead of RCCL, PT_GROUP_TRANSPORT_COPY; --transport forces either).
// --preview N (with --gpus): every N iterations the running average is converted on the devices and gathered
// (pt_group_preview_rgba8 — the reference shows it after every iteration, pathtrace.cu:618) into PREFIX.preview.png.
// Output name: PREFIX.<spp>samp.png, or with --stamp the reference's own
// <FILE>.<UTC start time>.<spp>samp.png (main.cpp:99-102).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/pathtrace_amd.hpp"
#include "pt_scene.h"

int main(int argc, char** argv) {
  if (argc < 2) {
    std::printf("Usage: %s SCENEFILE.txt [--res WxH] [--spp N] [--depth D] [--out PREFIX] [--pfm] [--hdr] "
                "[--arith exact|fma|fast] [--gpus K | --devices LIST] [--transport rccl|copy] [--stamp] [--aa] [--preview N]\n", argv[0]);
    return 1;
  }
  int rw = 0, rh = 0, spp = 0, depth = 0, gpus = -1, arith = PT_ARITH_EXACT, preview = 0, transport = PT_GROUP_TRANSPORT_AUTO;
  std::vector<int> device_list;
  bool pfm = false, hdr = false, stamp = false, aa = false;
  std::string out;
  for (int i = 2; i < argc; ++i) {
    if (!std::strcmp(argv[i], "--res") && i + 1 < argc) std::sscanf(argv[++i], "%dx%d", &rw, &rh);
    else if (!std::strcmp(argv[i], "--spp") && i + 1 < argc) spp = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "--depth") && i + 1 < argc) depth = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "--out") && i + 1 < argc) out = argv[++i];
    else if (!std::strcmp(argv[i], "--gpus") && i + 1 < argc) gpus = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "--preview") && i + 1 < argc) preview = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "--devices") && i + 1 < argc) {
      for (const char* q = argv[++i]; *q;) {
        char* end = nullptr;
        device_list.push_back((int)std::strtol(q, &end, 10));
        if (end == q) {
          std::fprintf(stderr, "--devices wants a comma-separated list of device ordinals\n");
          return 1;
        }
        q = *end == ',' ? end + 1 : end;
      }
      gpus = (int)device_list.size();
    } else if (!std::strcmp(argv[i], "--transport") && i + 1 < argc) {
      const char* a = argv[++i];
      if (!std::strcmp(a, "rccl")) transport = PT_GROUP_TRANSPORT_RCCL;
      else if (!std::strcmp(a, "copy")) transport = PT_GROUP_TRANSPORT_COPY;
      else {
        std::fprintf(stderr, "unknown transport %s\n", a);
        return 1;
      }
    }
    else if (!std::strcmp(argv[i], "--pfm")) pfm = true;
    else if (!std::strcmp(argv[i], "--hdr")) hdr = true;  // the Radiance file of image::saveHDR (main.cpp:106, commented out there)
    else if (!std::strcmp(argv[i], "--stamp")) stamp = true;
    else if (!std::strcmp(argv[i], "--aa")) aa = true;  // extension: stochastic anti-aliasing (PtOptions.aa_jitter)
    else if (!std::strcmp(argv[i], "--arith") && i + 1 < argc) {
      const char* a = argv[++i];
      if (!std::strcmp(a, "exact")) arith = PT_ARITH_EXACT;
      else if (!std::strcmp(a, "fma")) arith = PT_ARITH_FMA;
      else if (!std::strcmp(a, "fast")) arith = PT_ARITH_FAST;
      else {
        std::fprintf(stderr, "unknown arithmetic mode %s\n", a);
        return 1;
      }
    } else {
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
  if (out.empty()) out = scene->state.imageName;
  std::string base = out + "." + std::to_string(iters) + "samp";
  if (stamp) {
    char buf[1024];
    pt_output_basename(out.c_str(), iters, buf, sizeof buf);
    base = buf;
  }

  double secs = 0;
  if (gpus < 0) {
    // the reference's call sequence (main.cpp:133-152) through the pathtrace.h shim
    GuiDataContainer gui;
    InitDataContainer(&gui);
    pathtraceFree();  // main.cpp:134 frees before the first init
    pathtraceSetArith(arith);
    pathtraceSetAntialias(aa);
    pathtraceInit(scene);
    const auto t0 = std::chrono::high_resolution_clock::now();
    for (int it = 1; it <= iters; ++it) pathtrace(nullptr, 0, it);  // main.cpp:138-149
    pathtraceSyncImage();
    secs = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
    std::printf("%dx%d, %d spp, depth %d: %.3f s, %.2f Msamples/s\n", W, H, iters, scene->state.traceDepth, secs,
                (double)W * H * iters / secs / 1e6);
    if (pt_save_png((base + ".png").c_str(), scene->state.image.data(), W, H, (float)iters) == 0)
      std::printf("Saved %s.png.\n", base.c_str());
    if (pfm && pt_save_pfm((base + ".pfm").c_str(), scene->state.image.data(), W, H, (float)iters) == 0)
      std::printf("Saved %s.pfm.\n", base.c_str());
    if (hdr && pt_save_hdr((base + ".hdr").c_str(), scene->state.image.data(), W, H, (float)iters) == 0)
      std::printf("Saved %s.hdr.\n", base.c_str());
    pathtraceFree();
  } else {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
      std::fprintf(stderr, "no HIP device\n");
      return 1;
    }
    if (gpus == 0) gpus = ndev;
    if (device_list.empty() && gpus > ndev) {
      std::fprintf(stderr, "--gpus %d but only %d device(s) visible\n", gpus, ndev);
      return 1;
    }
    std::vector<int> devices(gpus);
    for (int i = 0; i < gpus; ++i) devices[i] = device_list.empty() ? i : device_list[i];
    for (int d : devices)
      if (d < 0 || d >= ndev) {
        std::fprintf(stderr, "device %d named but only %d device(s) visible\n", d, ndev);
        return 1;
      }
    PtOptions opt{};
    opt.arith = arith;
    opt.aa_jitter = aa ? 1 : 0;
    const PtSceneDesc desc = scene->desc();
    PtGroup* grp = nullptr;
    if (pt_group_create_ex(&desc, &opt, devices.data(), gpus, transport, &grp)) {
      std::fprintf(stderr, "HIP error (pt_group_create): %s\n", pt_last_error());
      return EXIT_FAILURE;
    }
    std::vector<uint8_t> rgb8((size_t)W * H * 3);
    const auto t0 = std::chrono::high_resolution_clock::now();
    int rc = 0;
    if (preview > 0) {
      std::vector<uint8_t> rgba((size_t)W * H * 4), rgb((size_t)W * H * 3);
      for (int it = 1; it <= iters && !rc; it += preview) {
        const int n = std::min(preview, iters - it + 1);
        rc = pt_group_render(grp, it, n);
        if (!rc && it + n <= iters) {  // progressive preview of what has been accumulated so far
          rc = pt_group_preview_rgba8(grp, it + n - 1, rgba.data());
          for (size_t p = 0; p < (size_t)W * H; ++p) rgb[3 * p] = rgba[4 * p], rgb[3 * p + 1] = rgba[4 * p + 1], rgb[3 * p + 2] = rgba[4 * p + 2];
          if (!rc) pt_write_png_rgb8((out + ".preview.png").c_str(), rgb.data(), W, H);
        }
      }
    } else {
      rc = pt_group_render(grp, 1, iters);
    }
    if (!rc) rc = pt_group_gather_u8(grp, (float)iters, rgb8.data());  // the write-out gather: 3 B per pixel
    secs = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
    if (rc) {
      std::fprintf(stderr, "HIP error (pt_group): %s\n", pt_last_error());
      return EXIT_FAILURE;
    }
    std::printf("%dx%d, %d spp, depth %d on %d GPU(s): %.3f s, %.2f Msamples/s\n", W, H, iters, scene->state.traceDepth, gpus,
                secs, (double)W * H * iters / secs / 1e6);
    if (pt_write_png_rgb8((base + ".png").c_str(), rgb8.data(), W, H) == 0) std::printf("Saved %s.png.\n", base.c_str());
    if (pfm || hdr) {
      scene->state.image.resize((size_t)W * H * 3);
      if (pt_group_gather(grp, scene->state.image.data())) {
        std::fprintf(stderr, "HIP error (pt_group_gather): %s\n", pt_last_error());
        return EXIT_FAILURE;
      }
      if (pfm && pt_save_pfm((base + ".pfm").c_str(), scene->state.image.data(), W, H, (float)iters) == 0)
        std::printf("Saved %s.pfm.\n", base.c_str());
      if (hdr && pt_save_hdr((base + ".hdr").c_str(), scene->state.image.data(), W, H, (float)iters) == 0)
        std::printf("Saved %s.hdr.\n", base.c_str());
    }
    pt_group_destroy(grp);
  }
  delete scene;
  return 0;
}
