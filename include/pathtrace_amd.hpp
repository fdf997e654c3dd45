// pathtrace_amd.hpp — C++ source-compatible mirror of the reference's renderer API
// (reference: src/pathtrace.h:6-9) on top of the C ABI in pt_amd.h.
//
//   reference                                   here
//   void InitDataContainer(GuiDataContainer*)   same
//   void pathtraceInit(Scene*)                  same, Scene = pt::Scene (pt_scene.h; same public
//                                               members geoms / materials / state as src/scene.h:20-25)
//   void pathtraceFree()                        same (legal before Init and twice, main.cpp:134)
//   void pathtrace(uchar4* pbo,int frame,int i) same; pbo may be nullptr (headless) or a DEVICE
//                                               pointer to W*H uchar4 (as the GL-mapped PBO is)
// Errors keep the reference's convention: message on stderr + exit(EXIT_FAILURE)
// (pathtrace.cu:141-150).
//
// One deliberate difference: the reference copies the whole accumulation image to
// scene->state.image after EVERY iteration (pathtrace.cu:648-651).  Here iterations
// are queued and traced in batches; scene->state.image is refreshed when the last
// iteration (state.iterations) has been submitted, on pathtraceFree(), or on demand
// with pathtraceSyncImage().  The image content is identical (a pure sum).
#pragma once
#include "pt_amd.h"

namespace pt {
class Scene;
}

class GuiDataContainer {  // src/utilities.h:17-22
 public:
  GuiDataContainer() : TracedDepth(0) {}
  int TracedDepth;
};

struct pt_uchar4 {
  unsigned char x, y, z, w;
};

void InitDataContainer(GuiDataContainer* guiData);
void pathtraceInit(pt::Scene* scene);
void pathtraceFree();
void pathtrace(pt_uchar4* pbo, int frame, int iteration);
// extension: flush queued iterations and refresh scene->state.image (running SUM)
void pathtraceSyncImage();
// extension: arithmetic mode (PT_ARITH_*, pt_amd.h) of the next pathtraceInit; default PT_ARITH_EXACT
void pathtraceSetArith(int pt_arith);
// extension: stochastic anti-aliasing (PtOptions.aa_jitter) of the next pathtraceInit; default off = reference semantics
void pathtraceSetAntialias(int on);
