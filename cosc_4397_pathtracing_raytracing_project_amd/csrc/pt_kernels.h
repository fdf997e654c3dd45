// pt_kernels.h — launch interface of the HIP kernels (pt_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "pt_device.h"

namespace ptk {

constexpr int kBlock = 256;        // 4 wave64 per workgroup
constexpr int kWavesPerBlock = 4;
constexpr int kLdsTableBytes = 64 * 1024;  // upper limit for staging the scene tables in LDS (see auto_lds_table_limit)
constexpr int kMaxTop = 32;                // entries in the flattened BVH top (per-lane 32-bit subtree mask)
constexpr int kCandCap = 192;              // per-wave candidate list entries (LDS)
constexpr int kWaveLds = 64 * 8 + 7 * 64 * 4 + kCandCap * 4;  // best keys + winner records + list = 3072 B

struct SceneTables {
  const ptd::Node* nodes;  // threaded DFS order
  int32_t num_nodes;
  const ptd::Geom* geoms;
  int32_t num_geoms;
  const ptd::Mat* mats;
  int32_t num_mats;
  const ptd::TopEntry* top;  // flattened BVH top (see ptd::TopEntry)
  int32_t num_top;
  float root_min[3], root_max[3];  // bounds of the whole tree (reference node 0)
  // Closer-hit cull of the subtree scans: a box whose entry distance exceeds the ray's best hit distance so far
  // by more than this margin cannot contain the closest hit.  The reported hit distance is measured to a point
  // pulled 1e-4 object units towards the ray origin (intersections.h:27-29) and carries the rounding of two
  // matrix products, so it can undershoot the true distance by 1e-4 * |transform| + O(1e-6 * |coordinates|); the
  // host sets a margin an order of magnitude above that bound (pt_api.cpp).  Negative: culling disabled.
  float cull_margin;
  // Near-first subtree order: byte k = XOR mask for rays whose direction sign bits are k = sx | sy << 1 | sz << 2
  // (pt_kernels.hip permute_xor); 0 when the top list is not a complete level of the tree.
  unsigned long long top_xor;
  // Host-side decision (KernelApi::auto_lds_table_limit): nodes + geoms up to this many bytes are staged in LDS by
  // the traversal kernels; -1: never.  Part of the tables so that several renderer contexts can differ.
  int32_t lds_table_bytes;
  // Iterations per wavefront batch of the context (>= every BatchInfo::K it launches): sizes the per-iteration RNG hash
  // table in LDS (none beyond 256 iterations, those batches hash per ray).
  int32_t max_batch_iters;
  // Uniform grid over the leaf boxes (pt_api.cpp build_grid; grid_search in pt_kernels.hip): the fused kernels of large
  // scenes walk it instead of the BVH when that is faster (measured by the host at init).  A cell lists every leaf whose box,
  // grown by grid_pad, overlaps it: cell c's records are grid_items[grid_start[c] .. grid_start[c + 1]), each a ptd::Node
  // with the leaf's box, `skip` = the leaf's threaded node index and `geom` = geom index << 8 | primitive type << 6 | bit a
  // set when the leaf is also listed in the neighbour cell a (0..5 = -x, +x, -y, +y, -z, +z).  A leaf is a candidate exactly when the ray passes its own box
  // test (pt_device.h, TopEntry), so any structure that finds a superset of those leaves gives the same image.
  const uint32_t* grid_start;
  const ptd::Node* grid_items;
  int32_t grid_res[3];
  float grid_min[3], grid_cs[3], grid_inv_cs[3];
  float grid_pad;
  int32_t use_grid;
  int32_t trace_depth;    // of the context (k_paths sizes its iteration-hash rows with it)
  int32_t has_triangles;  // mesh extension: some geoms are PT_GEOM_TRIANGLE (the sphere-only chunk specialisation is off)
  // The box tables of the BOUNCE kernels (depths >= 1; depth 0 always reads the ones above with the reference's arithmetic).  The
  // same tables, except in the fast build (KernelApi::boxes_center_half), whose slab test takes a box as centre and half extent
  // (pt_arith.inc slab_t): there the host uploads converted copies — bmin = centre, bmax = half extent rounded up so that the
  // box contains the original, inner nodes and subtree entries a little more (pt_api.cpp center_half_boxes).
  const ptd::Node* nodes_b;
  const ptd::TopEntry* top_b;
  const ptd::Node* grid_items_b;
  int32_t scan_nodes_lds;  // k_paths mode 1: 1 = the threaded nodes are staged in LDS for the subtree scans (scenes of a few hundred nodes:
                           // 64 primitives +4 % Msamples/s), 0 = read from memory, -1 = staged when that costs no resident workgroup per CU (resolved at launch)
};

struct BatchInfo {
  int32_t iter_first;   // iteration number (1-based) of sample plane k = 0
  int32_t K;            // iterations in this batch
  int32_t N;            // pixels in the tile
  int32_t pixel_begin;  // global index of tile pixel 0
  // Striped tiles (multi-GPU load balance): the tile is every `stripe`-pixel run out of `stripe + gap`;
  // tile pixel p is global pixel pixel_begin + p + (p / stripe) * gap.  stripe == 0: contiguous tile.
  int32_t stripe, gap;
  float inv_stripe;
  int32_t trace_depth;
  int32_t slot_shift;  // sample id of a path = k << slot_shift | tile pixel; 2^slot_shift >= N, k < 2^(31 - slot_shift)
  int32_t aa_jitter;  // 1: stochastic anti-aliasing of the camera rays (extension, PtOptions.aa_jitter)
  int32_t flat;   // 1: depth 0 appends its survivors to ONE dense depth-1 list per queue (the unfused k_intersect / k_shade pair
                  // reads that); 0: to one list per (queue, iteration) (k_paths, pt_device.h RetireBuf)
  int32_t primary_pieces;  // k_primary: a (queue, wave) strand's iterations cut into this many pieces, the first taken by the wave
                           // itself, the others by whoever is free (an atomic counter behind ptd::Queues::deal); <= 1: one piece
  int32_t paths_pieces;    // k_paths: low 16 bits: a queue's depth-1 rays cut into this many pieces per wave, one its own, the others first
                           // come, first served (counters behind ptd::Queues::deal); <= 1: one piece per wave; high 16 bits: fewest paths in a piece
  int32_t debug;  // profiling ablations (wrong results; honoured only by -DPT_ABLATE builds): 4 = skip primitive tests,
                  // 8 = skip shade_bounce
};

// Resident workgroups per CU for each persistent kernel (hipOccupancyMaxActiveBlocksPerMultiprocessor),
// so that grid = CUs * blocks never exceeds what is co-resident: work is dealt statically to waves,
// a workgroup that has to wait for a free slot would run its whole share after everybody else.
enum KernelId { kGenerate = 0, kIntersect = 1, kShade = 2, kIntersectLegacy = 3, kPrimary = 4, kPaths = 5 };

// pt_kernels.hip is compiled once per arithmetic mode (PtOptions.arith, include/pt_amd.h):
//   0 exact  -ffp-contract=off, every operation in the reference's order: bit-identical to oracle/pt_oracle.cpp
//            (PORTABLE mode) — the parity anchor;
//   1 fma    the same source with FMA contraction allowed (what nvcc does to the reference by default), IEEE
//            divide / sqrt kept, float-only direction sampling with accurate float sin / cos (ptmath::sincos_rev);
//   2 fast   fma + hardware reciprocal / rsqrt / sqrt / sin / cos (all <= 1 ulp or ~1e-6 absolute), nested-FMA
//            matrix products, float-only direction sampling.
// Modes 1 and 2 are checked against the oracle's reference semantics (LIBM mode) with the tolerance of
// SURVEY.md §8(c).  Each build exports its launch functions through this table.
struct KernelApi {
  const char* name;
  // generateRayFromCamera for all K*N samples of a batch, straight into the queues.
  // Also writes the queue fill counts cnt0[q*cnt_stride].
  void (*generate)(hipStream_t s, int grid, const ptd::Camera& cam, const BatchInfo& b, const ptd::Queues& qs,
                   ptd::PathBuf out, int32_t* cnt0);
  // Depth 0 fused (generate + intersect + shade + compaction): survivors go to `out` — one depth-1 list per (queue, iteration),
  // counted in `ret` (BatchInfo::flat: one list per queue, counted in cnt_out) — retired samples to the retirement records `ret`;
  // cnt0 receives the per-queue sample counts (statistics only).
  void (*primary)(hipStream_t s, int grid, const SceneTables& sc, const ptd::Camera& cam, const BatchInfo& b,
                  const ptd::Queues& qs, int32_t* cnt0, int32_t* cnt_out, ptd::PathBuf out, ptd::RetireBuf ret);
  // computeIntersections over the live paths of every queue.  exact_arith: these are primary rays (depth 0), which
  // are traced with the reference's exact arithmetic in every mode (pt_kernels.hip, namespace ex).
  void (*intersect)(hipStream_t s, int grid, const SceneTables& sc, const ptd::Queues& qs, const int32_t* cnt_in,
                    ptd::PathBuf paths, ptd::HitBuf hits, bool legacy, bool exact_arith);
  // shadeAndExtendRays + compaction + retirement.
  void (*shade)(hipStream_t s, int grid, const SceneTables& sc, const BatchInfo& b, int depth, const ptd::Queues& qs,
                const int32_t* cnt_in, int32_t* cnt_out, ptd::PathBuf in, ptd::HitBuf hits, ptd::PathBuf out,
                ptd::RetireBuf ret);
  // finalGather: image[p] += colour of sample (0, p) + (1, p) + ... in iteration order, from the retirement records
  // (ptd::RetireBuf); also resets the records' fill levels for the next batch.
  void (*collect)(hipStream_t s, const BatchInfo& b, const ptd::Queues& qs, ptd::RetireBuf ret, float* image_rgb /* [N][3] */);
  // live-ray bookkeeping: stats[d] += sum_q cnt[d][q]
  void (*count_stats)(hipStream_t s, const ptd::Queues& qs, int32_t* cnt, int depth_count, unsigned long long* stats);
  // sendImageToPBO (pathtrace.cu:250-268)
  void (*preview)(hipStream_t s, int n, int iterations, const float* image_rgb, uchar4* rgba);
  // saveImage + image::savePNG arithmetic on the device (main.cpp:86-107, image.cpp:22-39) for a tile of n pixels
  // made of whole rows: clamp(sum / samples, 0, 1) * 255 truncated, 3 bytes per pixel, x mirrored inside each row.
  void (*save_u8)(hipStream_t s, int n, int width, float samples, const float* image_rgb, uint8_t* rgb8);
  // Stage helper for tests: one shading step on n explicit paths (single queue, no compaction):
  // writes alive flags and in-place o/d/color.
  void (*shade_stage)(hipStream_t s, const SceneTables& sc, int trace_depth, int depth, int n, const int32_t* iter,
                      const int32_t* pixel, ptd::HitBuf hits, ptd::PathBuf paths, int32_t* alive);
  // Scene tables (nodes + geoms) up to the returned number of bytes are staged in LDS by the traversal kernels:
  // `forced_bytes` >= 0 is a test / experiment knob (only honoured for scenes whose leaves all fit the top list),
  // otherwise exactly when staging them costs the bounce kernel no resident block per CU.
  int (*lds_table_limit)(const SceneTables& sc, int forced_bytes);
  int (*resident_blocks_per_cu)(KernelId id, const SceneTables& sc);
  // Self-check of the guarded IEEE square root / reciprocal / quotient sequences (pt_kernels.hip namespace ieee) against the
  // compiler's expansions on `count` operands starting at `first`; adds the number of mismatching results to *bad.
  void (*ieee_check)(hipStream_t s, int kind, unsigned long long first, unsigned long long count, uint32_t seed, unsigned long long* bad);
  // ALL depths >= 1 of a batch in ONE launch — persistent lanes with their own depth, a dead lane takes the next depth-1 ray of its
  // queue, survivors never leave their registers (pt_kernels.hip k_paths; LDS-table scenes).  cnt = the counter rows [depth][Q]:
  // row 1 is read (the queues' depth-1 rays), rows >= 2 receive the rays traced per depth.
  void (*paths)(hipStream_t s, int grid, const SceneTables& sc, const BatchInfo& b, const ptd::Queues& qs, int32_t* cnt, ptd::PathBuf in, ptd::RetireBuf ret);
  int boxes_center_half;  // 1: the bounce kernels of this build expect SceneTables::*_b as centre / half extent (the fast build)
};
const KernelApi* api_exact();
const KernelApi* api_fma();
const KernelApi* api_fast();
inline const KernelApi* api_for(int arith) { return arith == 0 ? api_exact() : arith == 1 ? api_fma() : arith == 2 ? api_fast() : nullptr; }

}  // namespace ptk
