/* pt_amd.h — C ABI of the MI355X-native wavefront path tracer (libpt_amd.so).
 *
 * This is the drop-in boundary for the reference's renderer API
 * (reference: src/pathtrace.h:6-9 — InitDataContainer / pathtraceInit /
 * pathtraceFree / pathtrace) and for the host-side scene model it consumes
 * (src/scene.h:20-25, src/sceneStructs.h).  Plain C: pointers, sizes, PODs.
 * No C++ `Scene`, no GLM and no torch types cross this boundary.
 * The C++ source-compatible mirror of pathtrace.h lives in
 * include/pathtrace_amd.hpp; INTEGRATION.md shows the reference-side glue.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on error; pt_last_error()
 *     gives the message (the reference prints + exit()s, pathtrace.cu:141-150).
 *   - matrices are column-major float[16], element [c*4+r] == glm m[c][r].
 *   - images are float RGB, running SUM over iterations (not the average), in the
 *     reference's raw orientation (index = x + y*W; saveImage() mirrors x later,
 *     src/main.cpp:91-97), exactly what pathtrace() leaves in
 *     scene->state.image (pathtrace.cu:648-651).
 *   - single caller thread.  pt_init / pt_render / pt_free act on one default renderer instance, like the
 *     reference's file-scope state (pathtrace.cu:446-456); pt_ctx_* are the same operations on explicit
 *     instances (one per GPU), and pt_group_* drive several GPUs from one process with one RCCL gather at
 *     image write-out.
 */
#ifndef PT_AMD_H
#define PT_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PT_GEOM_SPHERE 0 /* sceneStructs.h:10-13 */
#define PT_GEOM_CUBE 1
/* EXTENSION (default scenes never contain it): one triangle of a `mesh` object — the type the scene format names
 * (INSTRUCTION.md:246) and the reference does not implement.  transform[0..8] holds the three WORLD-space vertices
 * v0.xyz, v1.xyz, v2.xyz, every other matrix element is 0; intersected as glm::intersectRayTriangle does (front faces
 * only; the header the reference includes at intersections.h:4).  Parity unpinned: tested GPU == oracle. */
#define PT_GEOM_TRIANGLE 2

/* The fields of `Geom` the renderer reads (sceneStructs.h:20-36). */
typedef struct PtGeom {
  int32_t type;
  int32_t materialid;
  float transform[16];
  float inverseTransform[16];
  float invTranspose[16];
} PtGeom;

/* Same field order and size (44 B) as `Material` (sceneStructs.h:38-48). */
typedef struct PtMaterial {
  float color[3];
  float specular_exponent;
  float specular_color[3];
  float hasReflective;
  float hasRefractive;
  float indexOfRefraction;
  float emittance;
} PtMaterial;

/* Same field order and size (84 B) as `Camera` (sceneStructs.h:50-59). */
typedef struct PtCamera {
  int32_t resolution[2];
  float position[3];
  float lookAt[3];
  float view[3];
  float up[3];
  float right[3];
  float fov[2];
  float pixelLength[2];
} PtCamera;

/* Same layout (36 B) as BVHNodeGPU (pathtrace.cu:24-32); for inspection/tests. */
typedef struct PtBVHNode {
  float bmin[3];
  float bmax[3];
  int32_t left, right, geomIndex;
} PtBVHNode;

typedef struct PtSceneDesc {
  const PtGeom* geoms;
  int32_t num_geoms;
  const PtMaterial* materials;
  int32_t num_materials;
  PtCamera camera;     /* after the main.cpp camera fix-up (see pt_scene_load) */
  int32_t trace_depth; /* RenderState::traceDepth */
} PtSceneDesc;

typedef struct PtOptions {
  int32_t device;          /* HIP device ordinal */
  int32_t pixel_begin;     /* framebuffer tile: global pixel indices             */
  int32_t pixel_count;     /*   [pixel_begin, pixel_begin+pixel_count); 0 = all  */
  int32_t iters_per_batch; /* iterations in flight per wavefront batch; 0 = auto */
  int32_t num_queues;      /* compaction queues; 0 = auto                        */
  int32_t blocks_per_cu;   /* persistent grid = CUs * blocks_per_cu; 0 = auto    */
  int32_t time_kernels;    /* 1: bracket every computeIntersections launch with  */
                           /*    HIP events on the render stream (pt_get_stats)  */
  int32_t legacy_traversal; /* 1: per-lane BVH walk kernel instead of the wave-cooperative one (A/B) */
  int32_t debug_flags;      /* A-B switches with UNCHANGED results: 16 no closer-hit cull in the subtree scans, 32 no
                               near-first subtree order, 256 / 512 force / forbid the uniform-grid walk of the fused kernels
                               (default: for large scenes, whichever of the BVH scan and up to three grid resolutions renders a
                               few iterations fastest at pt_init), 2048 keep the reference's leaf boxes for spheres (default for
                               large scenes: tightened to the ellipsoid's box, PtStats.tight_leaves; pt_stage_intersect on such
                               a scene then expects ray origins inside the scene bounds or at the camera).  Bits 0-3 are
                               profiling ablations with WRONG results (1 no top list, 4 skip the primitive tests, 8 skip the
                               bounce-direction sampling); they exist only in -DPT_ABLATE builds of the library
                               (tools/pmc_ablate.sh) and pt_init fails on them otherwise (pt_library_has_ablations()).
                               (Environment, tests only: PT_LDS_TABLE_KB forces the LDS staging limit of the scene tables.) */
  int32_t unfused_primary;  /* 1: run depth 0 as generate + intersect + shade launches instead of the fused
                               primary kernel (A/B and stage-parity runs) */
  int32_t unfused_bounces;  /* 1: depths >= 1 as separate computeIntersections + shade launches (hit records
                               through HBM) instead of the fused bounce kernel (A/B and stage-parity runs) */
  int32_t stripe_pixels;    /* striped tile for multi-GPU load balance: the tile consists of runs of          */
  int32_t stripe_stride;    /*   stripe_pixels pixels starting every stripe_stride pixels from pixel_begin;     */
                            /*   pixel_count counts the tile's own pixels.  0 = one contiguous run              */
  int32_t arith;            /* arithmetic mode of the kernels, PT_ARITH_*                                      */
  int32_t aa_jitter;        /* EXTENSION, default 0 = reference semantics.  1: stochastic anti-aliasing — the camera
                               ray of sample (iteration, pixel) goes through (x + u1 - .5, y + u2 - .5) instead of
                               the pixel centre; the reference's generateRayFromCamera ignores `iter`
                               (pathtrace.cu:270-286) although its assignment text asks for this (INSTRUCTION.md:96).
                               u1, u2 come from a hash domain of their own, every other random stream is unchanged.
                               Parity unpinned (nothing in the reference to compare with): tested GPU == oracle. */
  int32_t reserved[1];
} PtOptions;

/* Arithmetic modes (PtOptions.arith).  All modes run the same algorithm with the same random draws and decisions;
 * they differ in how float expressions are rounded.
 *   EXACT  every operation in the reference's source order without FMA contraction, IEEE divide / sqrt, portable
 *          sin / cos / acos: bit-identical to the CPU oracle (oracle/pt_oracle.cpp, PORTABLE mode).  The parity anchor.
 *   FMA    the same source with FMA contraction (what nvcc does to the reference's kernels by default), IEEE
 *          divide / sqrt kept; direction sampling with float-only sin / cos (<= 1.6 ulp, the accuracy class of the
 *          sinf / cosf nvcc links; the diffuse lobe through square roots instead of acos).
 *   FAST   FMA + hardware reciprocal / rsqrt / sqrt / sin / cos, nested-FMA matrix products, float-only
 *          direction sampling.
 * FMA and FAST are held to the stated tolerance against the reference semantics (tests/test_gpu_arith.py):
 * finite; at <= 16 spp >= 99.8 % of pixels within 1e-5; PSNR >= 45 dB + 10 log10(spp / 8). */
#define PT_ARITH_EXACT 0
#define PT_ARITH_FMA 1
#define PT_ARITH_FAST 2

#define PT_MAX_DEPTH 64
typedef struct PtStats {
  int64_t samples;                    /* pixel-samples rendered since pt_init            */
  int64_t live_rays[PT_MAX_DEPTH];    /* rays traced by computeIntersections per depth   */
  int64_t intersect_launches;         /* timed computeIntersections launches             */
  double intersect_ms;                /* sum of their HIP-event durations (time_kernels) */
  double render_ms;                   /* HIP-event time of all pt_render calls           */
  int32_t num_cus, grid_blocks, num_queues, iters_per_batch;
  int64_t device_bytes;               /* device memory held by the renderer              */
  int32_t primary_fused;              /* 1: depth 0 ran in the fused primary kernel, so the timed
                                         computeIntersections launches cover depths >= 1 only   */
  int32_t bounces_fused;              /* 1: depths >= 1 ran in the fused bounce kernel; the timed launches
                                         (intersect_ms / intersect_launches) are then those kernels      */
  int32_t arith;                      /* PT_ARITH_* in use                                               */
  int32_t grid_cells;                 /* > 0: the fused kernels walk the uniform grid over the leaf boxes (large scenes on
                                         which it beat the BVH scan at pt_init) with this many cells; 0: the BVH         */
  int32_t tight_leaves;               /* sphere leaves whose traversal box was tightened from the reference's box of the
                                         transformed unit cube to the box of the ellipsoid (large scenes; same image)     */
  int32_t paths_waves;                /* 0: every queue has the same number of waves in the fused bounce kernel; otherwise
                                         fewest << 16 | most waves a queue gets in the next batch — dealt by the time the
                                         queues' waves took in the last one (small tiles; same image, PT_NO_DEAL=1 turns it off) */
} PtStats;

/* ---- scene loading (host).  Replaces `new Scene(file)` (src/main.cpp:45,
 * src/scene.cpp:7-188) plus the camera state main.cpp derives before the first
 * frame (main.cpp:57-71, 110-128).  res_w/res_h > 0 override the RES line,
 * recomputing fov/pixelLength as scene.cpp:133-140 does. */
typedef struct PtScene PtScene;
int pt_scene_load(const char* path, int res_w, int res_h, PtScene** out);
void pt_scene_free(PtScene* s);
int pt_scene_desc(const PtScene* s, PtSceneDesc* out); /* pointers valid until pt_scene_free */
int pt_scene_iterations(const PtScene* s);             /* CAMERA ITERATIONS */
const char* pt_scene_image_name(const PtScene* s);     /* CAMERA FILE */

/* BVH exactly as pathtraceInit builds it (pathtrace.cu:34-111, 483-489).
 * Returns node count (2n-1); writes up to `cap` nodes if `out` != NULL. */
int pt_build_bvh(const PtGeom* geoms, int num_geoms, PtBVHNode* out, int cap);

/* The traversal structure of our own for large scenes (the reference has only the median-split
 * BVH of pathtrace.cu:52-111): a uniform grid over the leaf boxes, walked by the depth-0 and depth >= 1 kernels instead of
 * the BVH.  The image is the same either way: a primitive is tested exactly when the ray passes the primitive's own box
 * test, and every leaf is listed in all cells its box, grown by `pad`, touches.  A scene is a CANDIDATE when it has
 * >= 600 BVH nodes and its lists stay moderate (at most 64 cell references per primitive; `forced` skips both
 * conditions, as PtOptions.debug_flags 256 does); for a candidate pt_init / pt_ctx_create time a few iterations of the
 * tile with the BVH scan, with this grid and with two finer ones (4 and 8 cells per primitive) and keep the fastest
 * (PtStats.grid_cells > 0: a grid, with that many cells).  The device walks the same structure over the boxes of
 * pt_traversal_boxes, its cell table padded with empty guard cells.  This function is host-only (no GPU needed).  Returns 1 and fills `info` for a candidate, 0 otherwise, -1 on error; cell c's records
 * are records[cell_start[c] .. cell_start[c + 1]), c = x + res[0] * (y + res[1] * z); either array may be NULL (sizes
 * are in `info`). */
typedef struct PtGridInfo {
  int32_t res[3];
  float origin[3], cell_size[3], pad;
  int32_t num_cells, num_records, num_leaves;
} PtGridInfo;
typedef struct PtGridRecord {
  float bmin[3], bmax[3]; /* the leaf's box (the reference's worldBounds)                                     */
  int32_t leaf;           /* index of the leaf in the reference's visiting order (threaded BVH)              */
  int32_t neighbours;     /* bits 0-5: the leaf is also listed in the neighbour cell -x, +x, -y, +y, -z, +z;
                             bits 6-7: primitive type; bits 8-31: index of the primitive in PtSceneDesc.geoms  */
} PtGridRecord;
int pt_build_grid(const PtGeom* geoms, int num_geoms, int forced, PtGridInfo* info, uint32_t* cell_start, PtGridRecord* records);

/* Host-only: the box our traversal structures test for each geom's leaf — the reference's leaf box (pathtrace.cu:36-50),
 * except for spheres of large scenes, where it is intersected with the box of the ellipsoid itself (grown by a bound on
 * what sphereIntersectionTest's float arithmetic, intersections.h:102-144, can still report as a hit for ray origins inside
 * the scene bounds or at `camera_position`; csrc/pt_api.cpp sphere_tight_box).  A ray that passes the tightened box passes the
 * reference's; a ray that passes only the reference's misses the sphere: same hits, fewer candidates.  boxes[g] = {min xyz, max xyz}.
 * Returns the number of tightened leaves (pt_init applies it from 64 BVH nodes on; PtOptions.debug_flags 2048 turns it off). */
int pt_traversal_boxes(const PtGeom* geoms, int num_geoms, const float camera_position[3], float* boxes);

/* Host-only: a traversal box as the FAST build's bounce kernels test it — centre and half extent instead of min / max, so that
 * the slab test is three FMAs per axis (csrc/pt_arith.inc slab_t; min / max issue at half the rate of an FMA on gfx950).  The
 * half extent is rounded up from the distance between the float centre and the farther face, so [c - h, c + h] contains
 * [lo, hi] in real arithmetic; `inner` != 0 (inner nodes, subtree entries of the top list: pure acceleration) adds 1e-5 of the
 * extent and of the coordinates, so that a ray passing a leaf's box in the test's float arithmetic passes every box above it.
 * Depth 0 and the exact / fma builds test the reference's min / max boxes with the reference's arithmetic. */
void pt_center_half_box(const float lo[3], const float hi[3], int inner, float center[3], float half_extent[3]);

/* transform / inverse / inverse-transpose of an OBJECT block's TRANS ROTAT SCALE
 * (trs[9]), as utilityCore::buildTransformationMatrix + glm::inverse +
 * glm::inverseTranspose compute them (src/utilities.cpp:64-72, src/scene.cpp:83-86). */
int pt_build_transform(const float* trs, float* transform, float* inverse, float* invTranspose);

/* ---- renderer (replaces src/pathtrace.h) -------------------------------- */
int pt_init(const PtSceneDesc* scene, const PtOptions* opt); /* pathtraceInit, pathtrace.cu:462 */
int pt_free(void);                                           /* pathtraceFree, pathtrace.cu:518; safe
                                                                before init and twice in a row */
/* Runs iterations iter_first .. iter_first+iter_count-1 (1-based, as main.cpp:141-145
 * passes them) and adds them to the accumulation image.  Equivalent to iter_count
 * calls of pathtrace(pbo=NULL, 0, iter) (pathtrace.cu:529).  Asynchronous on the
 * renderer's stream; pt_sync / pt_readback wait. */
int pt_render(int iter_first, int iter_count);
int pt_sync(void);
/* Tile SUM image → host (pixel_count*3 floats), pathtrace.cu:648-651. */
int pt_readback(float* rgb_sum_host);
/* Tile SUM image → caller-owned device buffer (pixel_count*3 floats) on the
 * renderer's device; used to feed the RCCL gather without a host hop. */
int pt_readback_device(void* rgb_sum_dev);
/* Display conversion of sendImageToPBO (pathtrace.cu:250-268): average, gamma 1/2.2,
 * clamp → RGBA8 into a host buffer of pixel_count*4 bytes. */
int pt_preview_rgba8(int iterations, uint8_t* rgba_host);
int pt_preview_rgba8_device(int iterations, void* rgba_dev); /* same, into a device buffer (the PBO) */
int pt_get_stats(PtStats* out);
int pt_reset_stats(void);
/* Restart the accumulation without giving up the renderer: the SUM image and the statistics are zeroed, every buffer
 * stays allocated (and warm).  What the reference does by pathtraceFree() + pathtraceInit() when the camera moves
 * (main.cpp:134-135), minus the teardown. */
int pt_clear(void);
const char* pt_last_error(void);
int pt_library_has_ablations(void); /* 1: -DPT_ABLATE build (debug_flags bits 0-3 honoured) */
/* Device self-check.  The exact and fma kernels (and depth 0 of every mode) take correctly rounded square roots,
 * reciprocals and quotients from instruction sequences shorter than the compiler's general expansions whenever every
 * lane's operands are in a range where the two are the same computation (csrc/pt_kernels.hip, namespace ieee).  This runs
 * both side by side on the GPU and counts results that differ in any bit: kind 0 sqrt(x), 1 1/x, 2 1/sqrt(x) over the
 * `count` bit patterns starting at `first` (2^32 patterns = every float); kind 3 a/b, 4 the shared-reciprocal forms over
 * `count` pseudo-random operand sets derived from `seed` and the set's index first + i.  `arith` selects the kernel build. */
int pt_selfcheck_ieee(int arith, int kind, uint64_t first, uint64_t count, uint32_t seed, uint64_t* mismatches);
/* saveImage()'s per-pixel conversion (main.cpp:91-97 x mirror, image.cpp:26-30 clamp * 255 truncated) on the
 * device: pixel_count*3 bytes, row-major, x mirrored inside each row; the tile must consist of whole rows.
 * Reads back 3 B per pixel instead of 12. */
int pt_save_u8(float samples, uint8_t* rgb8_host);

/* ---- explicit renderer instances: the operations above on a context of your own (one per GPU). ---- */
typedef struct PtContext PtContext;
int pt_ctx_create(const PtSceneDesc* scene, const PtOptions* opt, PtContext** out); /* nothing is left behind on failure */
int pt_ctx_destroy(PtContext* c);
int pt_ctx_render(PtContext* c, int iter_first, int iter_count);
int pt_ctx_sync(PtContext* c);
int pt_ctx_readback(PtContext* c, float* rgb_sum_host);
int pt_ctx_readback_device(PtContext* c, void* rgb_sum_dev);
int pt_ctx_save_u8(PtContext* c, float samples, uint8_t* rgb8_host);
int pt_ctx_save_u8_device(PtContext* c, float samples, const uint8_t** rgb8_dev); /* async on the context's stream */
int pt_ctx_preview_rgba8(PtContext* c, int iterations, uint8_t* rgba_host);
int pt_ctx_preview_rgba8_device(PtContext* c, int iterations, void* rgba_dev);
int pt_ctx_get_stats(PtContext* c, PtStats* out);
int pt_ctx_reset_stats(PtContext* c);
int pt_ctx_clear(PtContext* c);
const float* pt_ctx_device_image(PtContext* c); /* device pointer of the tile SUM image */
void* pt_ctx_stream(PtContext* c);              /* the context's hipStream_t */
int pt_ctx_pixel_count(const PtContext* c);
int pt_ctx_device(const PtContext* c);

/* ---- one process, several GPUs (BASELINE config 4).  The framebuffer is cut into row-interleaved
 * tiles (device i of n owns rows i, i+n, ...; RNG keyed by the GLOBAL pixel index, so the assembled image is
 * bit-identical to the single-GPU image), every device renders its tile on its own stream with no data-path
 * collective, and the tiles meet once, at image write-out: one grouped RCCL send/recv (ncclCommInitAll
 * communicator) into device devices[0], placed row by row there, one D2H copy.  Replaces the reference's single
 * device (src/preview.cpp:112 cudaGLSetGLDevice(0)) and its write-out point (src/main.cpp:86-107). */
typedef struct PtGroup PtGroup;
/* Transport of that one exchange.  RCCL: grouped ncclSend / ncclRecv (default for distinct devices).  COPY:
 * hipMemcpyPeerAsync / hipMemcpyAsync of every tile into the same receive buffer, ordered by events — chosen
 * automatically when the device list names a device more than once (several contexts on one GPU, e.g. {0, 0, 0}:
 * RCCL cannot put two ranks of a communicator on one device), which makes the whole multi-context path runnable on a
 * one-GPU machine; selectable explicitly as a fallback.  The assembled image is the same bit for bit. */
#define PT_GROUP_TRANSPORT_AUTO 0
#define PT_GROUP_TRANSPORT_RCCL 1
#define PT_GROUP_TRANSPORT_COPY 2
int pt_group_create_ex(const PtSceneDesc* scene, const PtOptions* base, const int* devices, int num_devices, int transport,
                       PtGroup** out);
int pt_group_transport(const PtGroup* g); /* the resolved transport (PT_GROUP_TRANSPORT_RCCL or _COPY) */
/* = pt_group_create_ex(..., PT_GROUP_TRANSPORT_AUTO, out) */
int pt_group_create(const PtSceneDesc* scene, const PtOptions* base, const int* devices, int num_devices, PtGroup** out);
int pt_group_destroy(PtGroup* g);
int pt_group_size(const PtGroup* g);
PtContext* pt_group_context(PtGroup* g, int i);
int pt_group_render(PtGroup* g, int iter_first, int iter_count); /* asynchronous on every device */
int pt_group_sync(PtGroup* g);
int pt_group_gather(PtGroup* g, float* rgb_sum_host);             /* W*H*3 floats, raw orientation */
int pt_group_gather_u8(PtGroup* g, float samples, uint8_t* rgb8_host); /* W*H*3 bytes as pt_save_u8, converted on each device */
/* Progressive preview of the running average (sendImageToPBO, pathtrace.cu:250-268, which the reference runs after every
 * iteration): W*H RGBA8 bytes, raw orientation, converted on each device, one exchange of 4 B per pixel. */
int pt_group_preview_rgba8(PtGroup* g, int iterations, uint8_t* rgba_host);

/* ---- stage-level entry points (same kernels, caller-supplied HOST arrays, SoA:
 * vec3 arrays are [3][n]).  Used by the parity tests; each uploads, launches the
 * production kernel, downloads. ----------------------------------------------- */
/* generateRayFromCamera (pathtrace.cu:270-286) for pixels [pix_begin, pix_begin+n). */
int pt_stage_generate(int pix_begin, int n, float* origin, float* dir);
/* computeIntersections (pathtrace.cu:288-333): closest hit per ray. */
int pt_stage_intersect(int n, const float* origin, const float* dir, float* t, float* normal, int32_t* material,
                       float* point);
/* shadeAndExtendRays (pathtrace.cu:336-437) for live paths at `depth`:
 * in/out origin, dir, color; out alive[n] (1 = path continues).  Dead paths get the
 * retirement colour (sky factor applied (trace_depth - depth) times on a miss). */
int pt_stage_shade(int n, int depth, const int32_t* iter, const int32_t* pixel, const float* t, const float* normal,
                   const int32_t* material, const float* point, float* origin, float* dir, float* color,
                   int32_t* alive);

/* saveImage + savePNG conversion kernel (pt_save_u8) on a caller-supplied SUM image of w*h pixels (tests). */
int pt_stage_save_u8(int w, int h, float samples, const float* rgb_sum, uint8_t* rgb8);

/* ---- image output (src/image.cpp:22-45, src/main.cpp:86-107) ------------- */
/* rgb_sum: W*H*3 floats (raw orientation); writes <path> as 8-bit PNG of
 * clamp(sum/samples)*255 with the x mirror of saveImage(); no gamma. */
int pt_save_png(const char* path, const float* rgb_sum, int w, int h, float samples);
/* The same file from bytes that are already converted and mirrored (pt_save_u8 / pt_group_gather_u8). */
int pt_write_png_rgb8(const char* path, const uint8_t* rgb8, int w, int h);
/* "<name>.<UTC yyyy-mm-dd_hh-mm-ssz>.<samples>samp": the base name saveImage() composes (main.cpp:99-102,
 * preview.cpp:18-24 currentTimeString, taken once per process like main.cpp:35; `samples` is streamed as a float like
 * the reference's); writes at most cap bytes incl. the terminator, returns the full length. */
int pt_output_basename(const char* name, int samples, char* out, int cap);
int pt_save_pfm(const char* path, const float* rgb_sum, int w, int h, float samples);
/* image::saveHDR (src/image.cpp:41-45, stbi_write_hdr; main.cpp:106 keeps the call commented out) with saveImage()'s x
 * mirror and division by `samples`: Radiance RGBE, run-length coded, byte for byte what the reference's writer emits
 * (tests/golden/ref_hdr.json, made by the reference's image.cpp + stb.cpp compiled in place). */
int pt_save_hdr(const char* path, const float* rgb_sum, int w, int h, float samples);

#ifdef __cplusplus
}
#endif
#endif /* PT_AMD_H */
