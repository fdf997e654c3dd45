// pathtrace_shim.cpp — the four reference entry points (src/pathtrace.h:6-9)
// implemented over the C ABI.  See include/pathtrace_amd.hpp for the contract.
#include <cstdio>
#include <cstdlib>

#include "../../include/pathtrace_amd.hpp"
#include "pt_scene.h"

extern "C" int pt_preview_rgba8_device(int iterations, void* rgba_dev);

namespace {
pt::Scene* hst_scene = nullptr;
GuiDataContainer* guiData = nullptr;
int q_first = 0, q_count = 0;  // queued, not yet submitted iterations [q_first, q_first+q_count)
int last_iter = 0;
int arith_mode = PT_ARITH_EXACT;
int aa_mode = 0;

void check(int rc, const char* what) {  // pathtrace.cu:141-150
  if (rc == 0) return;
  std::fprintf(stderr, "HIP error (%s): %s\n", what, pt_last_error());
  std::exit(EXIT_FAILURE);
}
void flush() {
  if (q_count > 0) check(pt_render(q_first, q_count), "pathtrace");
  q_count = 0;
}
}  // namespace

void InitDataContainer(GuiDataContainer* imGuiData) { guiData = imGuiData; }
void pathtraceSetArith(int pt_arith) { arith_mode = pt_arith; }
void pathtraceSetAntialias(int on) { aa_mode = on ? 1 : 0; }

void pathtraceInit(pt::Scene* scene) {
  hst_scene = scene;
  PtSceneDesc d = scene->desc();
  PtOptions opt{};
  opt.arith = arith_mode;
  opt.aa_jitter = aa_mode;
  check(pt_init(&d, &opt), "pathtraceInit");
  q_count = 0;
  last_iter = 0;
}

void pathtraceSyncImage() {
  if (!hst_scene) return;
  flush();
  const PtCamera& c = hst_scene->state.camera;
  hst_scene->state.image.resize((size_t)c.resolution[0] * c.resolution[1] * 3);
  check(pt_readback(hst_scene->state.image.data()), "pathtraceSyncImage");
}

void pathtraceFree() {
  if (hst_scene && last_iter > 0) pathtraceSyncImage();
  check(pt_free(), "pathtraceFree");
  hst_scene = nullptr;
  q_count = 0;
  last_iter = 0;
}

void pathtrace(pt_uchar4* pbo, int /*frame*/, int iter) {
  if (!hst_scene) {
    std::fprintf(stderr, "pathtrace() before pathtraceInit()\n");
    std::exit(EXIT_FAILURE);
  }
  if (q_count > 0 && iter != q_first + q_count) flush();
  if (q_count == 0) q_first = iter;
  ++q_count;
  last_iter = iter;
  if (guiData) guiData->TracedDepth = hst_scene->state.traceDepth;
  const bool final_iter = iter >= (int)hst_scene->state.iterations;
  if (pbo) {
    flush();
    check(pt_preview_rgba8_device(iter, pbo), "sendImageToPBO");
  }
  if (final_iter || q_count >= 64) flush();
  if (final_iter) pathtraceSyncImage();
}
