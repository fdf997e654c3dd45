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
// 40 B per path, the algorithmic minimum of SURVEY §8(d) (+ 0: the id replaces the reference's pixelIndex).  -DPT_REC_TAGGED
// widens plane 2 to 16 B with utilhash(global pixel index) and k carried along (saves a hash and a shift per bounce; measured
// 10-20 % SLOWER — the bounce kernel's time follows its bytes — so it is not the default).
struct alignas(16) Word4 {
  float x, y, z, w;
};
#ifndef PT_REC_TAGGED
#define PT_REC_TAGGED 0
#endif
constexpr int kPathPlane2Bytes = PT_REC_TAGGED ? 16 : 8;  // plane 2 without the carried hash / iteration: colour.z, slot
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

// Retirement records.  A sample's final colour is known when its path dies — at any depth, in whatever order compaction
// has left the paths in — while finalGather (pathtrace.cu:439-444) needs, per pixel, the colours of its K samples in
// iteration order.  Rounds 1-2 stored each colour at final[k*N + p]: one scattered 16-byte write per sample, each a
// partially written DRAM line — a third of the bounce kernel's time at K = 25 (round 3 ablation).  Now:
//   * a queue owns the SAME pixel chunks in every iteration (Queues, below), i.e. a fixed set of nq*64 pixels;
//   * a retiring lane appends the record (r, g, b, tile pixel index) to a segment that belongs to (queue q, iteration k,
//     wave r of the queue's waves) alone — consecutive lanes to consecutive addresses, no atomics: the segment's fill
//     level is a counter private to that wave (kept in LDS during a kernel, in `cnt` between kernels);
//   * k_collect, one workgroup per queue, reads the segments of (q, k) for k = 0, 1, ... (coalesced), drops the colours
//     into an LDS tile indexed by pixel, and adds the tile to the queue's pixels — the reference's summation order.
// Every sample retires exactly once, so segment (q, k, *) fill levels add up to the queue's pixel count; a segment can
// hold all of them (seg_cap = nq * 64).
struct RetireBuf {
  Word4* rec;       // [Q][R][kmax][seg_cap]
  int32_t* cnt;     // [Q][R][kmax] fill levels (zero between batches: k_collect resets what it consumed)
  int32_t seg_cap;  // nq * 64: pixels a queue owns
  int32_t R;        // waves per queue the segments are provisioned for (>= W / Q of every launch)
  int32_t kmax;     // iterations per batch the segments are provisioned for
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
};

}  // namespace ptd
