// pt_device.h — device-side data layout of the MI355X wavefront path tracer.
//
// Everything the kernels stream is structure-of-arrays — planes of 16-byte words for the path state (PathBuf), 4-byte
// scalars for the hit records of the unfused form — so that a wave64 access is one contiguous run per plane (the
// reference's PathSegment is 44-B AoS and ShadeableIntersection 40-B AoS, src/sceneStructs.h:69-83).  Scene tables are
// tiny, read-only, and staged into LDS.
#pragma once
#include <stdint.h>

namespace ptd {

// BVH node, 32 B = two ds_read_b128.  The reference stores 36-B nodes with explicit
// left/right links and walks them with a 64-entry per-thread stack, pushing left
// then right, i.e. visiting right-subtree first (src/pathtrace.cu:28-32,302-324).
// That visiting order does not depend on the ray, so the tree is re-emitted in
// exactly that order (node, right subtree, left subtree) with a "skip" link to the
// first node after the subtree: traversal is then a stackless forward scan that
// performs the same node tests and the same leaf tests in the same order.
struct Node {
  float bmin[3];
  float bmax[3];
  int32_t skip;  // index of the next node when this subtree is not entered
  int32_t geom;  // >= 0: leaf, index into the geom table; -1: inner node
};
static_assert(sizeof(Node) == 32, "Node must be 32 B");

// Entry of the flattened BVH top ("top list").  A leaf is a candidate for a ray exactly when
// the ray passes the leaf's own AABB test: ancestors' boxes are unions of their children and
// the slab arithmetic is monotone, so an ancestor can never reject a ray its descendant
// accepts.  Inner nodes are therefore pure acceleration, and the top of the tree is replaced
// by a short list every ray tests with wave-uniform (scalar-loaded) box data; only subtrees
// below the cut are walked per lane.  For cornell.txt the cut is the 7 leaves themselves.
struct TopEntry {
  float bmin[3];
  float bmax[3];
  int32_t idx;   // threaded index of the leaf / subtree root (leaf: also the first-found order key)
  int32_t link;  // >= 0: subtree, one past its last threaded node;  < 0: leaf of type (-1 - link)
};
static_assert(sizeof(TopEntry) == 32, "TopEntry must be 32 B");

// Geometry record, 272 B.  Only rows 0..2 of each matrix are ever used
// (multiplyMV returns vec3, src/intersections.h:34-36), stored m[c*3+r] == glm m[c][r].
struct Geom {
  float inv[12];   // inverseTransform
  float xf[12];    // transform
  float invT[12];  // invTranspose
  int32_t type;    // 0 sphere, 1 cube (sceneStructs.h:10-13); 2 triangle (mesh extension): inv[0..8] = world-space v0, v1, v2
  int32_t material;
  int32_t pad[2];
  // Cube only: the world-space normal for each of the 7 values the object-space normal of
  // boxIntersectionTest can take (zero vector, -x, +x, -y, +y, -z, +z), i.e.
  // normalize(vec3(invTranspose * vec4(n, 0))) (intersections.h:86) evaluated on the host with the
  // same float operations in the same order — the kernel looks the result up instead of redoing a
  // mat*vec, a dot, a sqrt and a divide per candidate.  [code][xyz, pad]
  float box_normal[7][4];
};
static_assert(sizeof(Geom) == 272, "Geom must be 272 B");

// The five Material fields shading reads (sceneStructs.h:38-48), 48 B.
struct Mat {
  float color[3];
  float spec[3];
  float reflective;
  float refractive;
  float emittance;
  float pad[3];
};
static_assert(sizeof(Mat) == 48, "Mat must be 48 B");

struct Camera {
  int32_t res_x, res_y;
  float pos[3], view[3], up[3], right[3];
  float pl_x, pl_y;
};

// Ray / path state: three planes of 16-byte words, a path occupying the same index in each, so that a wave64 access is
// one contiguous 1-KB run per plane and a path costs three loads / three stores (round 1-2: ten 4-byte planes, i.e. ten
// memory instructions and ten 64-bit address computations each way; the kernels are bound by instruction issue):
//   plane 0   origin.xyz, direction.x
//   plane 1   direction.yz, colour.xy
//   plane 2   colour.z | sample id inside the batch: k << slot_shift | tile pixel (BatchInfo::slot_shift), 8 B per path
// 40 B per path, the algorithmic minimum of SURVEY §8(d) (+ 0: the id replaces the reference's pixelIndex).
struct alignas(16) Word4 {
  float x, y, z, w;
};
constexpr int kPathPlane2Bytes = 8;  // plane 2: colour.z, sample id
struct PathBuf {
  Word4* r;        // planes 0 and 1 at r, r + stride; plane 2 (kPathPlane2Bytes per path) at r + 2 * stride
  int64_t stride;  // paths per plane
};
// Hit records (the 32 live bytes of ShadeableIntersection).
struct HitBuf {
  float* t;        // [stride]
  float* n;        // [3][stride]
  int32_t* mat;    // [stride]
  float* p;        // [3][stride]
  int64_t stride;
};

// Retirement records.  A sample's final colour is known when its path dies — at any depth, in whatever order the lanes
// happen to finish — while finalGather (pathtrace.cu:439-444) needs, per pixel, the colours of its K samples in iteration
// order.  Rounds 1-2 stored each colour at final[k*N + p]: one scattered 16-byte write per sample, each a partially written
// DRAM line — a third of the bounce kernel's time at K = 25 (round 3 ablation).  Round 3 appended records to segments private
// to (queue, iteration, wave), each sized for ALL of the queue's pixels: coalesced, but 32 x over-provisioned (27 GB for a
// 1080p frame).  Now the fit is exact and nothing is reserved with atomics:
//   * a queue owns the SAME pixel chunks in every iteration (Queues, below): region (q, k) has one record slot per pixel of
//     queue q, seg_cap = nq * 64 of them, and the depth-1 list (q, k) — the survivors of depth 0 — lives at path index
//     q * cap + k * seg_cap, also seg_cap long;
//   * depth 0 (k_primary): in iteration k wave r of the queue's wq0 waves traces the chunks jj = rho, rho + wq0, ... with
//     rho = (r + k) mod wq0 (the residue rotates so that chunk counts even out over a batch).  Those are c(rho) chunks, and
//     off(rho) chunks belong to smaller residues, so SUB-list / SUB-region (q, k, rho) = slots [off(rho) * 64, (off(rho) +
//     c(rho)) * 64) of list / region (q, k) is the wave's own: it appends survivors to the sub-list and retirees to the front
//     of the sub-region from counters in its registers, stores at once, and leaves the two counts in sub[q][k][rho];
//   * depths >= 1 (k_paths) run after depth 0 of the whole batch.  A wave takes a contiguous slice of the queue's concatenated
//     sub-lists; the record of the path at index i of sub-list (k, rho) goes to slot retirees(k, rho) + i of sub-region
//     (k, rho): decided when the path is taken, no counters at all.  Every sample of the sub-region's chunks retires exactly
//     once, so the region ends up exactly full (but for the <= 63 slots of a tile's partial last chunk, which k_collect skips);
//   * k_collect, one workgroup per queue, reads region (q, k) for k = 0, 1, ... front to back, drops the colours into an LDS
//     tile indexed by pixel, and adds the tile to the queue's pixels — the reference's summation order.
// The unfused stage kernels (tests, A/B: BatchInfo::flat) append every record at the front of its region through cnt[q][k],
// one atomic per record, and keep ONE dense depth-1 list per queue.
struct RetireBuf {
  Word4* rec;               // [Q][kmax][seg_cap]
  unsigned long long* sub;  // [Q][kmax][wq0]: retirees << 32 | survivors of depth 0 in sub-region / sub-list (q, k, rho); rewritten by every batch
  unsigned long long* cnt;  // [Q][kmax]: flat form only — records appended at the front of region (q, k) << 32; zero between batches
  int32_t seg_cap;          // nq * 64: pixels a queue owns (a multiple of 64)
  int32_t kmax;             // iterations per batch the regions are provisioned for
  int32_t wq0;              // waves per queue of the k_primary launches (the sub-lists' residue count)
  int32_t pad;
};

// Work distribution.  Paths live in Q independent queues of capacity `cap`
// (queue q owns indices [q*cap, (q+1)*cap) of every plane).  The persistent grid has
// W waves; wave w serves queue w % Q together with the other W/Q - 1 waves of that
// queue, taking 64-path groups round-robin.  Survivors of a shading pass are appended
// to the same queue of the other PathBuf through one atomicAdd per wave on the
// queue's counter, so counters are spread over Q cache lines.
// Samples are dealt to the queues in 64-pixel chunks: chunk g of the tile (pixels 64 g .. 64 g + 63) belongs to queue
// g % Q in EVERY iteration (rounds 1-2 dealt the chunks of the whole batch round-robin, so a queue saw different pixels
// in every iteration); queue q's sample sequence is iteration-major: entry j = k * nq + jj is chunk q + jj * Q of
// iteration k, nq = ceil(chunks / Q).  A queue thus samples the frame on a regular lattice (every Q-th chunk), which
// balances the queues' work to within a few per cent without any dependence on the iteration.
struct Queues {
  int32_t Q;
  int32_t cap;
  int32_t W;            // total waves in the grid (multiple of Q)
  int32_t cnt_stride;   // ints between consecutive queue counters (64-B padding)
  int32_t nq;           // chunks per queue and iteration: ceil(ceil(N / 64) / Q)
  float inv_nq;         // 1.0f / nq (divmod)
  // k_paths' waves dealt to the queues by MEASURED work (null: W / Q each).  A queue owns the same pixels in every iteration, so
  // the rays its paths cost repeat from batch to batch, and with a small tile (a rank's share of a frame) they differ by
  // +-25 % between queues.  deal[0 .. Q] = first wave of queue q (deal[Q] = the W the table was made for: any other launch
  // width falls back to W / Q each), deal[Q + 1 + q] = time the waves of queue q spent in the k_paths launch of this batch (0.64-us
  // units) and deal[3 Q + 2 + q] = the rays they traced, from which k_count_stats deals the next batch's waves.  Which wave traces a path changes no sample (RetireBuf).  deal[2 Q + 1]: the
  // counter k_primary's waves take the later pieces of the strands from (BatchInfo::primary_pieces); deal[2 Q + 2 + q]: the counter
  // the waves of queue q take pieces of its depth-1 rays from in k_paths.  Both zero between batches (k_count_stats).
  int32_t* deal;
  int32_t paths_W;      // waves of the k_paths launches (what k_count_stats deals)
  int32_t pad;
};

}  // namespace ptd
