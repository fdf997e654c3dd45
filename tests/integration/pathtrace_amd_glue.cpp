// pathtrace_amd_glue.cpp — the reference-side binding of INTEGRATION.md §2, as a file that is actually compiled.
//
// A maintainer of the reference compiles THIS translation unit instead of src/pathtrace.cu and links libpt_amd.so:
// it implements the four entry points of src/pathtrace.h:6-9 against the reference's OWN headers (scene.h,
// sceneStructs.h, utilities.h, GLM) on top of the C ABI of include/pt_amd.h.  `make -C oracle ref` compiles it
// against /root/reference/src (and links it, with the reference's scene.cpp + utilities.cpp and
// tests/integration/glue_driver.cpp, into oracle/_ref/ref_glue_demo); tests/test_integration_glue.py checks that.
//
// Environment: PT_GLUE_ARITH = exact | fma | fast picks PtOptions.arith (default exact = bit-identical to the
// reference semantics).
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "scene.h"      // the reference's own Scene / Geom / Material / Camera (GLM types)
#include "pathtrace.h"  // the reference's own declarations (src/pathtrace.h:6-9)

#include "pt_amd.h"

// ---- layout contract: Material and Camera cross the boundary by memcpy (sceneStructs.h:38-59) ----
static_assert(sizeof(Material) == sizeof(PtMaterial) && sizeof(Material) == 44, "Material layout");
static_assert(offsetof(Material, color) == offsetof(PtMaterial, color), "Material.color");
static_assert(offsetof(Material, specular.exponent) == offsetof(PtMaterial, specular_exponent), "Material.specular.exponent");
static_assert(offsetof(Material, specular.color) == offsetof(PtMaterial, specular_color), "Material.specular.color");
static_assert(offsetof(Material, hasReflective) == offsetof(PtMaterial, hasReflective), "Material.hasReflective");
static_assert(offsetof(Material, hasRefractive) == offsetof(PtMaterial, hasRefractive), "Material.hasRefractive");
static_assert(offsetof(Material, indexOfRefraction) == offsetof(PtMaterial, indexOfRefraction), "Material.indexOfRefraction");
static_assert(offsetof(Material, emittance) == offsetof(PtMaterial, emittance), "Material.emittance");
static_assert(sizeof(Camera) == sizeof(PtCamera) && sizeof(Camera) == 84, "Camera layout");
static_assert(offsetof(Camera, resolution) == offsetof(PtCamera, resolution), "Camera.resolution");
static_assert(offsetof(Camera, position) == offsetof(PtCamera, position), "Camera.position");
static_assert(offsetof(Camera, lookAt) == offsetof(PtCamera, lookAt), "Camera.lookAt");
static_assert(offsetof(Camera, view) == offsetof(PtCamera, view), "Camera.view");
static_assert(offsetof(Camera, up) == offsetof(PtCamera, up), "Camera.up");
static_assert(offsetof(Camera, right) == offsetof(PtCamera, right), "Camera.right");
static_assert(offsetof(Camera, fov) == offsetof(PtCamera, fov), "Camera.fov");
static_assert(offsetof(Camera, pixelLength) == offsetof(PtCamera, pixelLength), "Camera.pixelLength");
static_assert(sizeof(glm::mat4) == 64 && sizeof(glm::vec3) == 12 && sizeof(glm::ivec2) == 8, "GLM types are packed");
static_assert((int)SPHERE == PT_GEOM_SPHERE && (int)CUBE == PT_GEOM_CUBE, "GeomType values (sceneStructs.h:10-13)");

static Scene* hst_scene = nullptr;
static GuiDataContainer* guiData = nullptr;

static void check(int rc, const char* what) {  // the reference's convention: print + exit (pathtrace.cu:141-150)
  if (rc) {
    fprintf(stderr, "HIP error (%s): %s\n", what, pt_last_error());
    exit(EXIT_FAILURE);
  }
}

void InitDataContainer(GuiDataContainer* imGuiData) { guiData = imGuiData; }

void pathtraceInit(Scene* scene) {
  hst_scene = scene;
  std::vector<PtGeom> geoms(scene->geoms.size());
  for (size_t i = 0; i < geoms.size(); ++i) {
    const Geom& g = scene->geoms[i];
    geoms[i].type = g.type;
    geoms[i].materialid = g.materialid;
    memcpy(geoms[i].transform, &g.transform[0][0], 64);  // glm::mat4 is column-major float[16]
    memcpy(geoms[i].inverseTransform, &g.inverseTransform[0][0], 64);
    memcpy(geoms[i].invTranspose, &g.invTranspose[0][0], 64);
  }
  PtSceneDesc d;
  memset(&d, 0, sizeof d);
  d.geoms = geoms.data();
  d.num_geoms = (int)geoms.size();
  d.materials = reinterpret_cast<const PtMaterial*>(scene->materials.data());
  d.num_materials = (int)scene->materials.size();
  memcpy(&d.camera, &scene->state.camera, sizeof(PtCamera));  // the camera runCuda() has already fixed up
  d.trace_depth = scene->state.traceDepth;
  PtOptions opt;
  memset(&opt, 0, sizeof opt);
  const char* a = getenv("PT_GLUE_ARITH");
  opt.arith = !a ? PT_ARITH_EXACT : !strcmp(a, "fast") ? PT_ARITH_FAST : !strcmp(a, "fma") ? PT_ARITH_FMA : PT_ARITH_EXACT;
  check(pt_init(&d, &opt), "pathtraceInit");  // copies everything it needs
}

void pathtraceFree() { check(pt_free(), "pathtraceFree"); }  // legal before any init and twice (main.cpp:134,152)

void pathtrace(uchar4* pbo, int /*frame*/, int iter) {
  check(pt_render(iter, 1), "pathtrace");  // asynchronous
  if (guiData) guiData->TracedDepth = hst_scene->state.traceDepth;
  if (pbo) check(pt_preview_rgba8_device(iter, pbo), "sendImageToPBO");  // pbo is a device pointer (the mapped GL buffer)
  // the reference refreshes state.image after every call (pathtrace.cu:648-651); do it when somebody can look at it
  if (iter >= (int)hst_scene->state.iterations || pbo)
    check(pt_readback(reinterpret_cast<float*>(hst_scene->state.image.data())), "readback");
}
