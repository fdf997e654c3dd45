// pt_kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the wavefront
// path tracer.  One persistent grid per stage; SoA path state; BVH + geometry +
// material tables staged in LDS; wave-level ballot/mbcnt compaction.
//
// What each kernel restates (reference paths relative to its repo root):
//   k_generate   generateRayFromCamera   src/pathtrace.cu:270-286
//   k_intersect  computeIntersections    src/pathtrace.cu:288-333, intersectAABB :113-128,
//                box/sphereIntersectionTest src/intersections.h:48-144
//   k_shade      shadeAndExtendRays      src/pathtrace.cu:336-437 (+ helpers :216-242,
//                RNG :203-207, utilhash intersections.h:12-20) fused with the retirement
//                rule of SURVEY.md §8a and the compaction the reference never had
//   k_collect    finalGather             src/pathtrace.cu:439-444 (sums the retirement records, pt_device.h RetireBuf,
//                into the image in iteration order)
//   k_preview    sendImageToPBO          src/pathtrace.cu:250-268
//   k_save_u8    saveImage + savePNG     src/main.cpp:86-107, src/image.cpp:22-39
// and the two kernels the default pipeline actually runs, which fuse the above so that neither the primary ray nor the
// hit record ever goes through HBM, and no path state after depth 0:
//   k_primary    depth 0:  generate + intersect + shade + compaction into one depth-1 list per (queue, iteration)
//   k_paths      ALL depths >= 1 in one launch: persistent lanes with their own depth, a dead lane takes the next
//                depth-1 ray; three search forms (LDS tables / top list + subtree scans / uniform grid walk)
// (k_generate / k_intersect / k_shade remain as the unfused form for stage-parity tests and A/B runs,
// k_intersect_legacy as the per-lane tree walk the wave-cooperative search replaced.)
//
// Arithmetic contract.  This file is compiled once per arithmetic mode (PT_ARITH, see pt_kernels.h KernelApi):
//   0 exact: -ffp-contract=off; every float operation is written in the order GLM 0.9.6 / the reference
//            evaluate it, divisions and square roots are IEEE (hipcc default
//            -fhip-fp32-correctly-rounded-divide-sqrt), and sin/cos/acos come from pt_portable_math.h, so
//            results are bit-identical to oracle/pt_oracle.cpp in PORTABLE mode;
//   1 fma:   the same source with contraction allowed; direction sampling in float only (shade_bounce_float<false>);
//   2 fast:  the `kFast` branches below — hardware rcp / rsq / sqrt / sin / cos, nested-FMA matrix products,
//            float-only direction sampling.  Same algorithm, same RNG draws, same decisions; only rounding differs.
// No MFMA: there is no dense contraction here.
#include "pt_kernels.h"

#include <float.h>

#include <type_traits>

#include "pt_portable_math.h"

#ifndef PT_ARITH
#define PT_ARITH 0
#endif
#if PT_ARITH == 0
#define PT_NS arith_exact
#define PT_API_FN api_exact
#elif PT_ARITH == 1
#define PT_NS arith_fma
#define PT_API_FN api_fma
#elif PT_ARITH == 2
#define PT_NS arith_fast
#define PT_API_FN api_fast
#else
#error "PT_ARITH must be 0, 1 or 2"
#endif

namespace ptk {
namespace PT_NS {
namespace {

#define PT_DEV __device__ __forceinline__

constexpr bool kFast = PT_ARITH == 2;
// Components of the fast mode, individually switchable (-DPT_FAST_x=0) for the flip-rate bisection of
// tools/arith_bisect.sh; the product builds leave all of them on.
#ifndef PT_FAST_TRIG
#define PT_FAST_TRIG 1  // direction sampling: v_sin / v_cos in revolutions, sqrt form of the diffuse lobe, no doubles
#endif
#ifndef PT_FAST_DIV
#define PT_FAST_DIV 1   // v_rcp_f32 instead of IEEE divides
#endif
#ifndef PT_FAST_SQRT
#define PT_FAST_SQRT 1  // v_rsq_f32 / v_sqrt_f32 instead of IEEE sqrt (+ divide) in normalize / length
#endif
#ifndef PT_FAST_SLAB
#define PT_FAST_SLAB 1  // AABB test as (b * inv - o * inv) FMAs with min/max per axis
#endif
#ifndef PT_FAST_MV
#define PT_FAST_MV 1    // nested-FMA matrix-vector products and dot products
#endif
#ifndef PT_FAST_QO
#define PT_FAST_QO 0    // 1: ray origin -> object space as a nested-FMA product too.  Off in the product: this is the one
                        // ill-conditioned product of the primitive test (thin wall: 100 * z + 500), and rounding it like the
                        // reference does (unfused, GLM order) removes 90 % of the remaining fma / fast sample flips
                        // (cornell 256^2 x 16 spp: 25 -> 3 pixels for fma, 46 -> 3 for fast; tools/arith_flips.py)
#endif
#ifndef PT_FAST_RENORM
#define PT_FAST_RENORM 1  // getPointOnRay does not normalise the already normalised object-space direction again
#endif
#ifndef PT_FAST_POINT
#define PT_FAST_POINT 0   // 1: hit point as ray origin + direction * world distance instead of transform * object-space point (the
                          // object-space direction is normalize(inverse * d), so transform * it = d / |inverse * d|, and the world
                          // distance is the object-space one times the rsq the normalisation already took): spares the 48-byte
                          // `transform` fetch of every hit candidate — a dependent second fetch per chunk — and 9 FMAs.  Measured
                          // (round 3): C5 bounce kernel -5.3 %, cornell -1 %; OFF because it is one more place where the fast
                          // mode rounds differently from the reference, and the random-scene tolerance test then sees 0.27 % of
                          // the pixels off by > 1e-5 against its bound of 0.2 % (tests/test_gpu_arith.py).  The tolerance wins.
#endif
constexpr bool kFastPoint = kFast && PT_FAST_POINT;
#ifndef PT_FMA_FLOAT_TRIG
#define PT_FMA_FLOAT_TRIG 1  // fma mode: float-only direction sampling (shade_bounce_float<false>) instead of the exact mode's
#endif
constexpr bool kFloatTrig = PT_ARITH == 1 && PT_FMA_FLOAT_TRIG;
constexpr bool kFastTrig = kFast && PT_FAST_TRIG, kFastDiv = kFast && PT_FAST_DIV, kFastSqrt = kFast && PT_FAST_SQRT,
               kFastSlab = kFast && PT_FAST_SLAB, kFastMV = kFast && PT_FAST_MV, kFastRenorm = kFast && PT_FAST_RENORM, kFastQO = PT_ARITH == 0 || (kFast && PT_FAST_QO);
// Ablation switches of tools/pmc_ablate.sh (BatchInfo::debug, wrong results) exist only in -DPT_ABLATE builds.
#ifdef PT_ABLATE
constexpr bool kAblate = true;
#else
constexpr bool kAblate = false;
#endif

// Diagnostic builds only (-DPT_WALK_STATS, tools/walk_stats.py): event counts of the large-scene kernels' search loops, added up
// over a launch in a buffer nothing else reads.  The product build compiles none of it.
#ifdef PT_WALK_STATS
__device__ unsigned long long g_walk_stats[16];
#define PT_STAT(i, v)                                                             \
  do {                                                                            \
    const unsigned long long pt_stat_v = (unsigned long long)(v); /* by every lane: v may hold a ballot */ \
    if (lane_id() == 0) atomicAdd(&g_walk_stats[i], pt_stat_v);                   \
  } while (0)
PT_DEV int wave_max_stat(int v) {
  for (int o = 32; o; o >>= 1) v = max(v, __shfl_xor(v, o));
  return v;
}
#else
#define PT_STAT(i, v)
#endif

// Tuning switch of the large-scene (global-table) path, overridable with -D for A/B builds.
#ifndef PT_STEAL_MIN
#define PT_STEAL_MIN 16
#endif
constexpr int kStealMin = PT_STEAL_MIN;  // idle lanes needed before a work-stealing step is run (65: never)
// Waves per SIMD the depth-0 kernel is compiled for (__launch_bounds__ second argument).  5: the camera ring's 5.9 KB per wave
// fit five workgroups per CU; in-box, Msamples/s at 4 | 5 | 6: fast 22.5 k | 22.5 k (95 VGPRs either way) | 21.0 k (88 B / lane of
// scratch), fma 17.27 k | 17.45 k, exact 15.29 k | 15.63 k (97 VGPRs at 4; 95 + 20 B / lane of scratch at 5).
#ifndef PT_PRIMARY_WAVES
#define PT_PRIMARY_WAVES 5
#endif
constexpr int kPrimaryWaves = PT_PRIMARY_WAVES;

typedef float v4f __attribute__((ext_vector_type(4)));
struct f3 {
  float x, y, z;
};
// ballot() of the HIP headers goes through an integer compare (v_cndmask 0 / 1 + v_cmp per call when the predicate already
// sits in a scalar register pair); the builtin takes the predicate as it is.
PT_DEV unsigned long long ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
// mask * 2 + bit in ONE instruction: v_addc_co_u32 takes the predicate as its carry-in (instead of v_mov + v_cndmask + v_or
// per box test).  The bits end up in reverse order of the pushes.
PT_DEV uint32_t push_bit(uint32_t m, bool bit) {
  uint32_t r;
  unsigned long long carry_out;
  asm("v_addc_co_u32_e64 %0, %1, %2, %2, %3" : "=v"(r), "=s"(carry_out) : "v"(m), "s"(ballot(bit)));
  return r;
}
// number of set bits of m below this lane
PT_DEV int rank_in(unsigned long long m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0)); }
PT_DEV f3 mk(float x, float y, float z) { return f3{x, y, z}; }

#include "pt_ieee.inc"
// Exact a / n and a % n for 0 <= a < 2^30 and quotients below 2^15 (sample ids / tile pixels,
// pixel index / image width): float estimate + one correction step either way, ~10 VALU instead of
// the ~35 of a 32-bit integer division.  inv_n = 1.0f / n computed once per kernel.
PT_DEV void divmod(int a, int n, float inv_n, int& q, int& r) {
  int k = (int)((float)a * inv_n);
  int rem = a - k * n;
  if (rem < 0) {
    rem += n;
    --k;
  } else if (rem >= n) {
    rem -= n;
    ++k;
  }
  q = k;
  r = rem;
}

// intersectAABB's per-ray constants (see ray_inv in pt_arith.inc)
struct RayInv {
  float ix, iy, iz;
  bool sx, sy, sz;
  float nx, ny, nz;  // fast mode: -origin * reciprocal, so that (b - o) / d is one FMA per plane
};

}  // namespace

// The arithmetic, twice (see pt_arith.inc).  Primary rays are identical in every iteration of a pixel, so a rounding
// difference that flips one of their decisions — the image diagonals of the square cornell frame look exactly along
// the box's corner edges, where two walls tie to the last bit of t — would repeat in every sample of that pixel instead
// of averaging out; measured: 90 % of all fma-vs-reference sample flips at 256 x 256 are such depth-0 ties.  Depth 0
// therefore runs the reference's exact primitive tests (namespace ex) in every arithmetic mode; the stochastic
// bounces use the mode's arithmetic (namespace md).
#pragma clang fp contract(off)
namespace ex {
namespace {
constexpr bool kFastDiv = false, kFastSqrt = false, kFastMV = false, kFastRenorm = false, kFastSlab = false, kFastQO = true, kFastPoint = false, kFusedRef = false;
#include "pt_arith.inc"
}  // namespace
}  // namespace ex
#if PT_ARITH != 0
#pragma clang fp contract(fast)
#endif
namespace md {
namespace {
constexpr bool kFastDiv = PT_NS::kFastDiv, kFastSqrt = PT_NS::kFastSqrt, kFastMV = PT_NS::kFastMV, kFastRenorm = PT_NS::kFastRenorm,
               kFastSlab = PT_NS::kFastSlab, kFastQO = PT_NS::kFastQO, kFastPoint = PT_NS::kFastPoint, kFusedRef = PT_ARITH == 1;
#include "pt_arith.inc"
}  // namespace
}  // namespace md
namespace {
using namespace md;
// Arithmetic flavour of a call site: Ar<true> = the reference's exact arithmetic (namespace ex), Ar<false> = the
// build's mode.  kD0 selects the flavour of everything geometric at depth 0; in the exact build both are the same
// arithmetic, so it stays on md and nothing is instantiated twice.
constexpr bool kD0 = PT_ARITH != 0;
template <bool EX>
struct Ar {
  static PT_DEV RayInv ray_inv(f3 d, f3 o) { if constexpr (EX) return ex::ray_inv(d, o); else return md::ray_inv(d, o); }
  static PT_DEV bool slab(f3 o, const RayInv& ri, float a, float b, float c, float d, float e, float f) {
    if constexpr (EX) return ex::slab(o, ri, a, b, c, d, e, f); else return md::slab(o, ri, a, b, c, d, e, f);
  }
  static PT_DEV bool slab_rel(const RayInv& ri, float a, float b, float c, float d, float e, float f) {
    if constexpr (EX) return ex::slab_rel(ri, a, b, c, d, e, f); else return md::slab_rel(ri, a, b, c, d, e, f);
  }
  static PT_DEV bool slab_t(f3 o, const RayInv& ri, float a, float b, float c, float d, float e, float f, float& tn) {
    if constexpr (EX) return ex::slab_t(o, ri, a, b, c, d, e, f, tn); else return md::slab_t(o, ri, a, b, c, d, e, f, tn);
  }
  template <int TYPE, bool QO, bool DEFER = false>
  static PT_DEV float geom_test(const ptd::Geom* __restrict__ G, f3 ro, f3 rd, f3& point, f3& normal, f3 qo_pre, int rt_type = -1) {
    if constexpr (EX) return ex::geom_test<TYPE, QO, DEFER>(G, ro, rd, point, normal, qo_pre, rt_type);
    else return md::geom_test<TYPE, QO, DEFER>(G, ro, rd, point, normal, qo_pre, rt_type);
  }
  static PT_DEV f3 finish_normal(const ptd::Geom* __restrict__ G, f3 stored) {
    if constexpr (EX) return ex::finish_normal(G, stored); else return md::finish_normal(G, stored);
  }
  static PT_DEV f3 camera_dir(const ptd::Camera& cam, float inv_w, int p, bool aa, float jx, float jy) {
    if constexpr (EX) return ex::camera_dir(cam, inv_w, p, aa, jx, jy); else return md::camera_dir(cam, inv_w, p, aa, jx, jy);
  }
  static PT_DEV f3 xf_point(const float* m, f3 v) { if constexpr (EX) return ex::mulMV<1>(m, v); else return md::mulMV<1>(m, v); }
};

PT_DEV int lane_id() { return (int)(threadIdx.x & 63); }
// value of v in lane src_lane
PT_DEV float bperm(int src_lane, float v) {
  return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane << 2, __float_as_int(v)));
}

// ───────────────────────────── RNG ──────────────────────────────────────────
PT_DEV uint32_t utilhash(uint32_t a) {  // intersections.h:12-20
  a = (a + 0x7ed55d16u) + (a << 12);
  a = (a ^ 0xc761c23cu) ^ (a >> 19);
  a = (a + 0x165667b1u) + (a << 5);
  a = (a + 0xd3a2646cu) ^ (a << 9);
  a = (a + 0xfd7046c5u) + (a << 3);
  a = (a ^ 0xb55a4f09u) ^ (a >> 16);
  return a;
}
// thrust::minstd_rand (a = 48271, m = 2^31 - 1) with Mersenne folding instead of %.
struct MinStd {
  uint32_t x;
  PT_DEV explicit MinStd(uint32_t s) {
    uint32_t r = (s & 0x7fffffffu) + (s >> 31);  // s mod (2^31-1)
    if (r >= 0x7fffffffu) r -= 0x7fffffffu;
    x = r == 0u ? 1u : r;
  }
  PT_DEV uint32_t next() {
    uint64_t p = (uint64_t)x * 48271u;  // < 2^47
    uint32_t r = (uint32_t)(p & 0x7fffffffu) + (uint32_t)(p >> 31);
    if (r >= 0x7fffffffu) r -= 0x7fffffffu;
    x = r;
    return r;
  }
  // uniform_real_distribution<float>(0,1): float(x - 1) / 2^31 (exact scaling)
  PT_DEV float u01() { return (float)(next() - 1u) * 4.656612873077392578125e-10f; }
};
// makeSeededRandomEngine (pathtrace.cu:203-207): seed = utilhash((1 << 31) | (depth << 22) | iter) ^ utilhash(index).
// The first factor depends only on (iteration, depth): the kernels compute it once per iteration of the batch into
// a small LDS table instead of once per ray (same values, ~18 VALU less per ray).
PT_DEV uint32_t iter_hash(int iter, int depth) { return utilhash((1u << 31) | ((uint32_t)depth << 22) | (uint32_t)iter); }
constexpr int kIterHashMax = 256;  // most table entries (iterations per batch it covers; larger batches hash per ray)
// LDS entries of the table: the context's iterations per batch (SceneTables::max_batch_iters), none beyond kIterHashMax
__host__ __device__ inline int iter_hash_entries(const SceneTables& sc) {
  return sc.max_batch_iters <= kIterHashMax ? (sc.max_batch_iters + 3) & ~3 : 0;
}
PT_DEV void iter_hash_fill(uint32_t* tab, const SceneTables& sc, const BatchInfo& b, int depth) {  // before the kernel's __syncthreads()
  if (iter_hash_entries(sc) > 0)
    for (int i = threadIdx.x; i < b.K; i += blockDim.x) tab[i] = iter_hash(b.iter_first + i, depth);
}
PT_DEV uint32_t iter_hash_of(const uint32_t* tab, const SceneTables& sc, const BatchInfo& b, int depth, int k) {
  return iter_hash_entries(sc) > 0 ? tab[k] : iter_hash(b.iter_first + k, depth);
}

// Anti-aliasing jitter of sample (iteration, global pixel): two draws of an engine seeded in a hash domain of its own
// ("depth" field 0x100: bit 30, which no path depth < 64 produces), so the streams of the reference semantics are untouched.
PT_DEV void aa_jitter(int iter, int pixel, float& jx, float& jy) {
  MinStd rng(utilhash((1u << 31) | (1u << 30) | (uint32_t)iter) ^ utilhash((uint32_t)pixel));
  jx = rng.u01() - 0.5f;
  jy = rng.u01() - 0.5f;
}

// ───────────────────────────── LDS staging ─────────────────────────────────
// Copies `bytes` (multiple of 16) from global to LDS with 16-B accesses.
PT_DEV void stage16(void* lds, const void* g, int bytes) {
  const float4* src = reinterpret_cast<const float4*>(g);
  float4* dst = reinterpret_cast<float4*>(lds);
  for (int i = threadIdx.x; i < bytes / 16; i += blockDim.x) dst[i] = src[i];
}

// Tile pixel → global pixel index (what keys the RNG and the camera ray), see BatchInfo::stripe.
PT_DEV int global_pixel(const BatchInfo& b, int p) {
  if (b.stripe == 0) return b.pixel_begin + p;
  int i, r;
  divmod(p, b.stripe, b.inv_stripe, i, r);
  return b.pixel_begin + p + i * b.gap;
}

// ───────────────────────────── path records (ptd::PathBuf) ─────────────────
// What rides along with a path besides its ray and colour: the sample id.  (phash / k: the per-pixel half of the RNG seed and the
// iteration inside the batch, kept in registers by the kernels that have them; they are not part of the record in memory — a
// round-3 build that carried them, 48 B per path, saved 56 VALU per 64-ray group and was 10-20 % slower.)
struct PathTag {
  int slot;
  uint32_t phash;  // utilhash(global pixel index)
  int k;           // iteration inside the batch: iteration = BatchInfo::iter_first + k
};
struct PathRec {
  f3 o, d, c;
  PathTag tag;
};
struct alignas(8) Word2 {
  float x, y;
};
// plane 2: colour.z, slot
PT_DEV void plane2_load(const ptd::PathBuf& b, int64_t at, float& cz, PathTag& tag) {
  const Word2 w2 = reinterpret_cast<const Word2*>(b.r + 2 * b.stride)[at];
  cz = w2.x;
  tag.slot = __float_as_int(w2.y), tag.phash = 0u, tag.k = 0;
}
PT_DEV void plane2_store(const ptd::PathBuf& b, int64_t at, float cz, const PathTag& tag) {
  reinterpret_cast<Word2*>(b.r + 2 * b.stride)[at] = Word2{cz, __int_as_float(tag.slot)};
}
PT_DEV void path_load_ray(const ptd::PathBuf& b, int64_t at, f3& o, f3& d) {  // planes 0, 1
  const ptd::Word4 w0 = b.r[at], w1 = b.r[b.stride + at];
  o = mk(w0.x, w0.y, w0.z);
  d = mk(w0.w, w1.x, w1.y);
}
PT_DEV void path_load_tail(const ptd::PathBuf& b, int64_t at, f3& d, f3& c, PathTag& tag) {  // planes 1, 2 + direction.x
  const ptd::Word4 w1 = b.r[b.stride + at];
  float cz;
  plane2_load(b, at, cz, tag);
  d = mk(reinterpret_cast<const float*>(b.r + at)[3], w1.x, w1.y);
  c = mk(w1.z, w1.w, cz);
}
PT_DEV PathRec path_load(const ptd::PathBuf& b, int64_t at) {
  const ptd::Word4 w0 = b.r[at], w1 = b.r[b.stride + at];
  PathRec v;
  float cz;
  plane2_load(b, at, cz, v.tag);
  v.o = mk(w0.x, w0.y, w0.z);
  v.d = mk(w0.w, w1.x, w1.y);
  v.c = mk(w1.z, w1.w, cz);
  return v;
}
PT_DEV void path_store(const ptd::PathBuf& b, int64_t at, f3 o, f3 d, f3 c, const PathTag& tag) {
  b.r[at] = ptd::Word4{o.x, o.y, o.z, d.x};
  b.r[b.stride + at] = ptd::Word4{d.y, d.z, c.x, c.y};
  plane2_store(b, at, c.z, tag);
}
// Iteration (inside the batch) and tile pixel of a path from its sample id k << slot_shift | pl (BatchInfo::slot_shift: a
// shift and a mask instead of the division by N that the id k * N + pl of rounds 1-2 needed at every depth).
PT_DEV int make_slot(const BatchInfo& b, int k, int pl) { return (k << b.slot_shift) | pl; }
PT_DEV void sample_of(const PathTag& tag, const BatchInfo& b, int& k, int& pl) {
  k = (int)((uint32_t)tag.slot >> b.slot_shift);
  pl = tag.slot & ((1 << b.slot_shift) - 1);
}
// makeSeededRandomEngine's seed (pathtrace.cu:205) of a path at `depth`: utilhash((1 << 31) | depth << 22 | iteration) ^
// utilhash(global pixel index).  The first factor comes from the per-block table (iter_hash_of).
PT_DEV uint32_t path_seed(int k, int pl, const uint32_t* ihash, const SceneTables& sc, const BatchInfo& b, int depth) {
  return iter_hash_of(ihash, sc, b, depth, k) ^ utilhash((uint32_t)global_pixel(b, pl));
}

// ───────────────────────────── retirement records (ptd::RetireBuf) ─────────
// One queue's regions and counters (pt_device.h RetireBuf).
struct Retire {
  ptd::Word4* rec;          // region (q, 0); region (q, k) starts k * seg_cap records further
  unsigned long long* sub;  // [kmax][wq0]: retirees << 32 | survivors of depth 0 per sub-region / sub-list (q, k, rho)
  unsigned long long* cnt;  // [kmax]: flat form — records appended at the front of region (q, k) << 32
  int seg_cap, wq0;
};
PT_DEV Retire retire_of(const ptd::RetireBuf& rb, int q) {
  Retire rt;
  rt.rec = rb.rec + (int64_t)q * rb.kmax * rb.seg_cap;
  rt.sub = rb.sub + (int64_t)q * rb.kmax * rb.wq0;
  rt.cnt = rb.cnt + (int64_t)q * rb.kmax;
  rt.seg_cap = rb.seg_cap, rt.wq0 = rb.wq0;
  return rt;
}
// Sub-list / sub-region (q, k, rho) of a queue with my_nq chunks per iteration dealt to wq0 residues: c(rho) chunks, the first
// of them off(rho) chunks into list / region (q, k)  (residue rho owns the chunks jj = rho, rho + wq0, ...).
// (quo = my_nq / wq0, rem = my_nq % wq0, computed once per kernel)
PT_DEV int sub_chunks(int quo, int rem, int rho) { return quo + (rho < rem ? 1 : 0); }
PT_DEV int sub_offset(int quo, int rem, int rho) { return rho * quo + min(rho, rem); }
PT_DEV void retire_store(const Retire& rt, bool dead, int k, int pos, int pl, f3 c) {
#ifdef PT_ABL_NO_RETIRE  // timing experiment only (wrong images)
  return;
#endif
  if (dead) rt.rec[(int64_t)k * rt.seg_cap + pos] = ptd::Word4{c.x, c.y, c.z, __int_as_float(pl)};
}
// The unfused stage kernels: every record appended at the front of its region, one returning atomic per record on the
// region's counter (test / A-B form; the fused kernels take one atomic per 64 samples or none at all).
PT_DEV void retire_append(const Retire& rt, bool dead, int k, int pl, f3 c) {
  if (dead) retire_store(rt, true, k, (int)(atomicAdd(&rt.cnt[k], 1ull << 32) >> 32), pl, c);
}
// Queue q's share of an iteration (ptd::Queues): chunks q, q + Q, ... of the tile's ceil(N / 64); only the tile's last
// chunk can be partial, and it is the last chunk of the queue that owns it.
struct QueueShare {
  int my_nq;      // chunks of this queue per iteration
  int my_pixels;  // pixels of this queue per iteration
  float inv_my_nq;
};
PT_DEV QueueShare queue_share(const BatchInfo& b, const ptd::Queues& qs, int q) {
  const int chunks = (b.N + 63) >> 6;
  QueueShare sh;
  sh.my_nq = q < chunks ? (chunks - q + qs.Q - 1) / qs.Q : 0;
  const int last_q = (chunks - 1) % qs.Q;
  sh.my_pixels = sh.my_nq * 64 - ((q == last_q && (b.N & 63)) ? 64 - (b.N & 63) : 0);
  sh.inv_my_nq = sh.my_nq > 0 ? 1.0f / (float)sh.my_nq : 0.0f;
  return sh;
}

// ───────────────────────────── generate ────────────────────────────────────
__global__ __launch_bounds__(kBlock) void k_generate(ptd::Camera cam, BatchInfo b, ptd::Queues qs, ptd::PathBuf out,
                                                     int32_t* __restrict__ cnt0) {
  const int wave = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int lane = lane_id();
  const int q = wave % qs.Q, r = wave / qs.Q, wq = qs.W / qs.Q;
  const float inv_w = 1.0f / (float)cam.res_x;
  const QueueShare sh = queue_share(b, qs, q);
  if (r == 0 && lane == 0) cnt0[(size_t)q * qs.cnt_stride] = b.K * sh.my_pixels;
  const int entries = b.K * sh.my_nq;  // iteration-major: entry j = k * my_nq + jj is chunk q + jj * Q of iteration k
  for (int j = r; j < entries; j += wq) {
    int k, jj;
    divmod(j, sh.my_nq, sh.inv_my_nq, k, jj);
    const int pl = (q + jj * qs.Q) * 64 + lane;  // tile pixel
    if (pl < b.N) {
      const int p = global_pixel(b, pl);  // global pixel index
      float jx = 0.f, jy = 0.f;
      if (b.aa_jitter) aa_jitter(b.iter_first + k, p, jx, jy);
      const f3 d = Ar<kD0>::camera_dir(cam, inv_w, p, b.aa_jitter != 0, jx, jy);
      const int64_t at = (int64_t)q * qs.cap + (int64_t)k * sh.my_pixels + jj * 64 + lane;  // dense: a partial chunk is the queue's last
      path_store(out, at, mk(cam.pos[0], cam.pos[1], cam.pos[2]), d, mk(1.0f, 1.0f, 1.0f), PathTag{make_slot(b, k, pl), utilhash((uint32_t)p), k});
    }
  }
}

// ───────────────────────────── intersect ───────────────────────────────────
struct HitRec {
  float t;
  f3 n;
  f3 p;
  int geom;
};

// Near-first order of a ray's entered subtrees.  When the top list is a complete level of a balanced tree its
// entries are laid out by path code (bit 4 = child taken at the root, ... bit 0 = at level 4; pt_api.cpp build_top),
// so visiting the nearer child first at every level is visiting the entries in increasing (index XOR m) order,
// where bit l of m says that the nearer child at that level is child 1 for this ray's direction signs
// (SceneTables::top_xor, one mask per sign octant).  permute_xor returns the pending mask re-indexed by
// j = e ^ m, so that ctz yields the entries nearest first; the cull then prunes the farther ones earlier.
// Any m is valid (the order never changes results); m = 0 keeps the list order.
PT_DEV uint32_t permute_xor(uint32_t x, uint32_t m) {
  if (m & 1u) x = ((x & 0x55555555u) << 1) | ((x >> 1) & 0x55555555u);
  if (m & 2u) x = ((x & 0x33333333u) << 2) | ((x >> 2) & 0x33333333u);
  if (m & 4u) x = ((x & 0x0f0f0f0fu) << 4) | ((x >> 4) & 0x0f0f0f0fu);
  if (m & 8u) x = ((x & 0x00ff00ffu) << 8) | ((x >> 8) & 0x00ff00ffu);
  if (m & 16u) x = (x << 16) | (x >> 16);
  return x;
}
PT_DEV uint32_t octant_mask(const RayInv& ri, unsigned long long top_xor) {
  const int oct = (int)ri.sx | ((int)ri.sy << 1) | ((int)ri.sz << 2);
  return (uint32_t)(top_xor >> (oct * 8)) & 31u;
}

// One step of a lane's stackless subtree scan.  On return `cur` is advanced, `cand` says whether the node is a
// leaf whose box the ray passes (and that is not culled), `leaf` is its index in `nodes`, `geom` its geom index.
// bt: the ray's best hit distance so far + SceneTables::cull_margin (closer-hit cull).
template <bool EX = false, typename NP = const v4f*>
PT_DEV void scan_step(NP nodes4, f3 o, const RayInv& ri, bool act, int& cur, float bt,
                      bool& cand, int& leaf, int& geom) {
  const int at_n = act ? cur : 0;
  const v4f NA = nodes4[2 * at_n];      // bmin.xyz, bmax.x
  const v4f NB = nodes4[2 * at_n + 1];  // bmax.yz, skip, geom
  float tn;
  const bool in = act && Ar<EX>::slab_t(o, ri, NA.x, NA.y, NA.z, NA.w, NB.x, NB.y, tn) && !(tn > bt);
  geom = __float_as_int(NB.w);
  cand = in && geom >= 0;
  leaf = at_n;
  if (act) cur = in ? cur + 1 : __float_as_int(NB.z);
}

// State of a lane's subtree scan: the subtree range being scanned and the ray it is scanned for (the lane's own
// ray, or a donor's after a steal — see steal_step).
struct Walker {
  int cur, end;  // next node / one past the subtree's last node (cur >= end: idle)
  int own;       // lane that owns the ray (candidates are filed under it)
  f3 o;          // ray origin
  RayInv ri;     // reciprocal direction
};
// Work stealing inside the wave.  The scan loop runs max-over-lanes steps, and lanes' totals range from 0 to several
// hundred node visits (rays enter 0-10 subtrees, a subtree costs 5-150 visits), so lanes whose own pending list
// is empty take one pending (ray, subtree) pair each from lanes that still have some: donors publish their lane id
// in `slot` (64 ints of LDS) by rank, the k-th idle lane reads the k-th donor's id, fetches its pending mask, XOR
// mask, ray origin and reciprocal direction with ds_bpermute, and the donor drops the pair it gave away.
// Called at wave-uniform control flow.  A lane only steals once its own list is empty, hence a donor's Walker
// always holds the donor's own ray.  pend: pending subtrees in permute_xor order.
PT_DEV void steal_step(Walker& w, uint32_t& pend, uint32_t xm, bool idle, unsigned long long I, int* slot,
                       const float4* top, int lane) {
  const unsigned long long Dn = ballot(pend != 0);
  if (!Dn) return;
  const int nd = __popcll(Dn), ni = __popcll(I);
  const int drank = __builtin_amdgcn_mbcnt_hi((uint32_t)(Dn >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)Dn, 0));
  const int irank = __builtin_amdgcn_mbcnt_hi((uint32_t)(I >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)I, 0));
  if (pend != 0) slot[drank] = lane;  // the k-th donor's lane id
  const bool take = idle && irank < nd;
  const int donor = slot[take ? irank : 0];
  const uint32_t dpend = (uint32_t)__builtin_amdgcn_ds_bpermute(donor << 2, (int)pend);
  const int dxm = __builtin_amdgcn_ds_bpermute(donor << 2, (int)xm);
  const f3 so = mk(bperm(donor, w.o.x), bperm(donor, w.o.y), bperm(donor, w.o.z));
  const f3 si = mk(bperm(donor, w.ri.ix), bperm(donor, w.ri.iy), bperm(donor, w.ri.iz));
  if (pend != 0 && drank < ni) pend &= pend - 1;  // given away
  if (take) {
    const int e = __builtin_ctz(dpend) ^ dxm;
    const float4 TB = top[2 * e + 1];
    w.cur = __float_as_int(TB.z) + 1;  // the subtree root's box is the top entry's box: already passed
    w.end = __float_as_int(TB.w);
    w.own = donor;
    w.o = so;
    w.ri.ix = si.x, w.ri.iy = si.y, w.ri.iz = si.z;
    w.ri.sx = si.x < 0.0f, w.ri.sy = si.y < 0.0f, w.ri.sz = si.z < 0.0f;
    w.ri.nx = -so.x * si.x, w.ri.ny = -so.y * si.y, w.ri.nz = -so.z * si.z;
  }
}
// Legacy traversal (kept for A/B measurements, PtOptions flag): one lane walks the threaded
// tree and runs each primitive test as soon as the wave reaches it.
template <bool EX>
PT_DEV HitRec trace(const ptd::Node* __restrict__ nodes, int num_nodes, const ptd::Geom* __restrict__ geoms, f3 o, f3 d) {
  HitRec h;
  h.t = FLT_MAX;
  h.geom = -1;
  h.n = mk(0.f, 0.f, 0.f);
  h.p = mk(0.f, 0.f, 0.f);
  const RayInv ri = Ar<EX>::ray_inv(d, o);
  int i = 0;
  while (true) {
    int g = -1;
    while (i < num_nodes) {
      const float4 A = reinterpret_cast<const float4*>(nodes)[2 * i];      // bmin.xyz, bmax.x
      const float4 B = reinterpret_cast<const float4*>(nodes)[2 * i + 1];  // bmax.yz, skip, geom
      if (!Ar<EX>::slab(o, ri, A.x, A.y, A.z, A.w, B.x, B.y)) {
        i = __float_as_int(B.z);
        continue;
      }
      i = i + 1;
      const int leaf = __float_as_int(B.w);
      if (leaf >= 0) {
        g = leaf;
        break;
      }
    }
    if (g < 0) break;
    f3 pt, nrm;
    const float t = Ar<EX>::template geom_test<-1, false>(geoms + g, o, d, pt, nrm, mk(0.f, 0.f, 0.f));
    if (t > 0.f && t < h.t) {  // strict <: first found wins ties (pathtrace.cu:314)
      h.t = t;
      h.geom = g;
      h.p = pt;
      h.n = nrm;
    }
  }
  return h;
}

template <bool TABLES_IN_LDS, bool EX>
__global__ __launch_bounds__(kBlock) void k_intersect_legacy(SceneTables sc, ptd::Queues qs,
                                                             const int32_t* __restrict__ cnt_in, ptd::PathBuf paths,
                                                             ptd::HitBuf hits) {
  extern __shared__ float4 lds_raw[];
  const ptd::Node* nodes = EX ? sc.nodes : sc.nodes_b;  // EX: primary rays, the reference's arithmetic on the reference's boxes; else the build's bounce tables
  const ptd::Geom* geoms = sc.geoms;
  if (TABLES_IN_LDS) {
    char* base = reinterpret_cast<char*>(lds_raw);
    stage16(base, nodes, sc.num_nodes * (int)sizeof(ptd::Node));
    stage16(base + sc.num_nodes * sizeof(ptd::Node), sc.geoms, sc.num_geoms * (int)sizeof(ptd::Geom));
    __syncthreads();
    nodes = reinterpret_cast<const ptd::Node*>(base);
    geoms = reinterpret_cast<const ptd::Geom*>(base + sc.num_nodes * sizeof(ptd::Node));
  }
  const int wave = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int lane = lane_id();
  const int q = wave % qs.Q, r = wave / qs.Q, wq = qs.W / qs.Q;
  const int n_q = cnt_in[(size_t)q * qs.cnt_stride];
  const int64_t HS = hits.stride;
  for (int j = r; j * 64 < n_q; j += wq) {
    const int i = j * 64 + lane;
    if (i < n_q) {
      const int64_t at = (int64_t)q * qs.cap + i;
      f3 o, d;
      path_load_ray(paths, at, o, d);
      const HitRec h = trace<EX>(nodes, sc.num_nodes, geoms, o, d);
      const bool hit = h.geom >= 0;
      // record layout of the reference after its per-depth memset (pathtrace.cu:562):
      // miss → t = -1 and zeros elsewhere.
      hits.t[at] = hit ? h.t : -1.0f;
      hits.n[at] = h.n.x, hits.n[HS + at] = h.n.y, hits.n[2 * HS + at] = h.n.z;
      hits.mat[at] = hit ? geoms[h.geom].material : 0;
      hits.p[at] = h.p.x, hits.p[HS + at] = h.p.y, hits.p[2 * HS + at] = h.p.z;
    }
  }
}

// ── computeIntersections, wave-cooperative form ──────────────────────────────────────────
// Per group of 64 rays a wave runs
//   phase 1  candidate search: every lane tests its ray against the flattened BVH top (box data
//            wave-uniform → scalar loads) and walks entered subtrees with the stackless scan;
//            each (ray, leaf) whose AABB test passes is appended to a per-wave LDS list with a
//            ballot + mbcnt prefix — cubes from the front, spheres from the back;
//   phase 2  primitive tests over the list in dense 64-entry chunks, so all lanes run the same
//            (type-specialised) code on useful work; the ray is fetched from its owner lane with
//            ds_bpermute; the closest hit per ray is kept with an LDS 64-bit atomic min on
//            (t bits << 32 | leaf index).  The reference takes a hit iff t > 0 && t < t_min in its
//            DFS visiting order (pathtrace.cu:314), i.e. the smallest t and, among equal t, the
//            leaf visited first; leaves are numbered in that order, so the key's minimum is
//            exactly the reference's choice no matter in which order candidates are evaluated.
// The reference has no t_min culling, so the candidate set does not depend on hit results and
// the two phases are independent.
struct WaveLds {
  unsigned long long* best;  // [64]
  float* rec;                // [7][64]: normal xyz, point xyz; row 6: donor table of the work-stealing step
  uint32_t* list;            // [kCandCap]: (leaf index << 6) | owner lane
};
constexpr unsigned long long kNoHit = ((unsigned long long)0x7f7fffffu << 32) | 0xffffffffull;  // t_min = FLT_MAX


// Runs the pending candidates; called at wave-uniform control flow with all 64 lanes active.
// One chunk of <= 64 candidates: `nc` cubes starting at list[cfirst] followed by `nsph` spheres starting at
// list[sfirst].  TYPE 1 / 0: the chunk holds only cubes / only spheres (specialised code); TYPE -1: both —
// the object-space transform of the ray and the world-space reconstruction are executed once for all
// lanes and only the slab / quadratic middle parts diverge (geom_test<-1>).
// CAM (primary kernel): every ray starts at the camera, so the origin needs no fetch; QO (tables in LDS): its
// object-space image comes from the per-geom table qo_tab.
template <int TYPE, bool CAM, bool QO, bool EX>
PT_DEV void run_chunk(const WaveLds& w, int cfirst, int nc, int sfirst, int nsph, int lane, f3 o, f3 d,
                      const ptd::Node* __restrict__ nodes, const ptd::Geom* __restrict__ geoms, const float* qo_tab) {
  const bool valid = lane < nc + nsph;
  const uint32_t entry = valid ? w.list[lane < nc ? cfirst + lane : sfirst + (lane - nc)] : (uint32_t)lane;
  const int src = (int)(entry & 63u);
  const uint32_t leaf = entry >> 6;
  const f3 ro = CAM ? o : mk(bperm(src, o.x), bperm(src, o.y), bperm(src, o.z));
  const f3 rd = mk(bperm(src, d.x), bperm(src, d.y), bperm(src, d.z));
  const int gi = valid ? nodes[leaf].geom : 0;
  const ptd::Geom* G = geoms + gi;
  f3 pt, nrm;
  float t;
  if (QO) t = Ar<EX>::template geom_test<TYPE, true>(G, ro, rd, pt, nrm, mk(qo_tab[3 * gi], qo_tab[3 * gi + 1], qo_tab[3 * gi + 2]));
  else t = Ar<EX>::template geom_test<TYPE, false>(G, ro, rd, pt, nrm, mk(0.f, 0.f, 0.f));
  const uint32_t tb = __float_as_uint(t);
  if (valid && t > 0.f && tb < 0x7f7fffffu) {
    const unsigned long long key = ((unsigned long long)tb << 32) | leaf;
    atomicMin(&w.best[src], key);
    if (w.best[src] == key) {  // this candidate is the ray's best so far: publish its record
      w.rec[0 * 64 + src] = nrm.x, w.rec[1 * 64 + src] = nrm.y, w.rec[2 * 64 + src] = nrm.z;
      w.rec[3 * 64 + src] = pt.x, w.rec[4 * 64 + src] = pt.y, w.rec[5 * 64 + src] = pt.z;
    }
  }
}
// Runs the pending candidates; called at wave-uniform control flow with all 64 lanes active.
// Chunk plan (a typical group at depth >= 1 holds ~55 cubes and ~12 spheres): full chunks of cubes,
// then the remaining cubes together with the spheres in ONE mixed chunk if they fit in 64 lanes
// (costs ~1.3x a pure chunk instead of two pure chunks), otherwise separately.
template <bool CAM, bool QO, bool EX>
PT_DEV void flush_candidates(const WaveLds& w, int nb, int ns, int lane, f3 o, f3 d,
                             const ptd::Node* __restrict__ nodes, const ptd::Geom* __restrict__ geoms, const float* qo_tab,
                             bool tri) {  // tri (wave-uniform): the back list may hold triangles (mesh extension)
  const int sbase = kCandCap - ns;
  int c0 = 0;
  for (; c0 + 64 <= nb; c0 += 64) run_chunk<1, CAM, QO, EX>(w, c0, 64, 0, 0, lane, o, d, nodes, geoms, qo_tab);
  const int rem = nb - c0;
  if (rem > 0 && ns > 0 && rem + ns <= 64) {
    run_chunk<-1, CAM, QO, EX>(w, c0, rem, sbase, ns, lane, o, d, nodes, geoms, qo_tab);
    return;
  }
  if (rem > 0) run_chunk<1, CAM, QO, EX>(w, c0, rem, 0, 0, lane, o, d, nodes, geoms, qo_tab);
  for (int s0 = 0; s0 < ns; s0 += 64) {
    if (tri) run_chunk<-1, CAM, QO, EX>(w, 0, 0, sbase + s0, min(64, ns - s0), lane, o, d, nodes, geoms, qo_tab);
    else run_chunk<0, CAM, QO, EX>(w, 0, 0, sbase + s0, min(64, ns - s0), lane, o, d, nodes, geoms, qo_tab);
  }
}

// Phase 1 + phase 2 for one group of 64 rays (one per lane; `valid` masks tail lanes).  On return
// w.best[lane] holds the lane's (t bits << 32 | leaf) key (kNoHit if none) and w.rec its normal/point.
// CAM: primary rays — `top` holds the entries' boxes relative to the camera position (slab_rel) and qo_tab the
// camera position in every geom's object space (run_chunk<.., true>).
template <bool CAM, bool QO, bool EX>
PT_DEV void trace_group(const WaveLds& w, const float4* top, int ntop, const ptd::Node* __restrict__ nodes,
                        const ptd::Geom* __restrict__ geoms, f3 o, f3 d, bool valid, int lane, float cull,
                        unsigned long long top_xor, const float* qo_tab = nullptr, bool tri = false) {
  const RayInv ri = Ar<EX>::ray_inv(d, o);
  w.best[lane] = kNoHit;
  int nb = 0, ns = 0;  // pending cubes (front of the list) / spheres (back)
  uint32_t pend = 0;   // per lane: top entries that are subtrees and whose box this ray passes

  // top list: wave-uniform LDS reads (broadcast), entry e+1 fetched while entry e is tested
  float4 A = top[0], B = top[1];
  for (int e = 0; e < ntop; ++e) {
    const float4 TA = A, TB = B;  // bmin.xyz, bmax.x | bmax.yz, idx, link
    if (e + 1 < ntop) A = top[2 * e + 2], B = top[2 * e + 3];
    const int t_idx = __builtin_amdgcn_readfirstlane(__float_as_int(TB.z));
    const int t_link = __builtin_amdgcn_readfirstlane(__float_as_int(TB.w));
    const bool pass = valid && (CAM ? Ar<EX>::slab_rel(ri, TA.x, TA.y, TA.z, TA.w, TB.x, TB.y) : Ar<EX>::slab(o, ri, TA.x, TA.y, TA.z, TA.w, TB.x, TB.y));
    if (t_link < 0) {  // leaf entry: type is wave-uniform
      const unsigned long long m = ballot(pass);
      if (m) {
        if (nb + ns + 64 > kCandCap) {
          flush_candidates<CAM, QO, EX>(w, nb, ns, lane, o, d, nodes, geoms, qo_tab, tri);
          nb = ns = 0;
        }
        const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
        const int cnt = __popcll(m);
        const uint32_t entry = ((uint32_t)t_idx << 6) | (uint32_t)lane;
        if (t_link == -2) {  // cube
          if (pass) w.list[nb + rank] = entry;
          nb += cnt;
        } else {
          if (pass) w.list[kCandCap - ns - cnt + rank] = entry;
          ns += cnt;
        }
      }
    } else if (pass) {
      pend |= 1u << e;
    }
  }
  // subtrees below the cut: every lane walks its entered subtrees back to back, independently of the others, nearest
  // first; lanes without work steal pending (ray, subtree) pairs (steal_step; the donor table lives in the spare row
  // of w.rec).  Candidates are filed under the owner's lane and the primitive tests fetch the ray from the owner's
  // registers o, d as before.
  if (ballot(pend != 0)) {
    Walker wk{0, 0, lane, o, ri};
    int* slot = reinterpret_cast<int*>(w.rec + 6 * 64);
    const uint32_t xm = octant_mask(ri, top_xor);
    pend = permute_xor(pend, xm);
    while (true) {
      if (wk.cur >= wk.end && pend) {  // own subtrees first, nearest first
        const int e = __builtin_ctz(pend) ^ (int)xm;
        pend &= pend - 1;
        const float4 TB = top[2 * e + 1];
        wk.cur = __float_as_int(TB.z) + 1;  // the subtree root's box is the top entry's box: already passed
        wk.end = __float_as_int(TB.w);
      }
      const bool idle = wk.cur >= wk.end;
      const unsigned long long I = ballot(idle);
      if (I == ~0ull) break;
      if (__popcll(I) >= kStealMin) steal_step(wk, pend, xm, idle, I, slot, top, lane);
      const bool act = wk.cur < wk.end;
      // closer-hit cull: a box entered beyond the ray's best hit so far (+ margin, see SceneTables::cull_margin)
      // cannot hold the closest hit
      const float bt = __uint_as_float(reinterpret_cast<const uint32_t*>(w.best)[2 * wk.own + 1]) + cull;
      bool cand;
      int at_n, aux;
      scan_step<EX>(reinterpret_cast<const v4f*>(nodes), wk.o, wk.ri, act, wk.cur, bt, cand, at_n, aux);
      const bool cbox = cand && geoms[aux].type == 1;
      const bool csph = cand && !cbox;
      const unsigned long long mb = ballot(cbox), msp = ballot(csph);
      if (mb | msp) {
        if (nb + ns + 128 > kCandCap) {
          flush_candidates<CAM, QO, EX>(w, nb, ns, lane, o, d, nodes, geoms, qo_tab, tri);
          nb = ns = 0;
        }
        const uint32_t entry = ((uint32_t)at_n << 6) | (uint32_t)wk.own;
        const int rb = __builtin_amdgcn_mbcnt_hi((uint32_t)(mb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mb, 0));
        const int rs = __builtin_amdgcn_mbcnt_hi((uint32_t)(msp >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)msp, 0));
        const int cb = __popcll(mb), cs = __popcll(msp);
        if (cbox) w.list[nb + rb] = entry;
        if (csph) w.list[kCandCap - ns - cs + rs] = entry;
        nb += cb;
        ns += cs;
      }
    }
  }
  if (nb + ns) flush_candidates<CAM, QO, EX>(w, nb, ns, lane, o, d, nodes, geoms, qo_tab, tri);
}

// Primary rays of a scene whose tables stay in memory (more than 32 leaves, no grid): the candidate search as ONE scan of the
// threaded tree for the whole group.  The 64 rays of a group start at the camera and go through neighbouring pixels, so they
// enter nearly the same subtrees: the wave walks the reference's visiting order with wave-uniform node data (scalar loads),
// every lane tests the node's box for its own ray, and a subtree is skipped when NO lane passes its root (the threaded
// `skip` link) — the nodes visited are the union of the lanes' own walks, ~1.5 walks' worth instead of 32 top-list tests plus a
// per-lane scan with work stealing.  Same candidates: a lane files a leaf exactly when its ray passes the leaf's own box, and
// whoever passes that passes every ancestor (exact unions, monotone slab arithmetic), so the lane would have reached it.
typedef __attribute__((address_space(4))) const v4f cnode4;  // wave-uniform indices into it become scalar loads
template <bool EX>
PT_DEV void trace_group_packet(const WaveLds& w, const ptd::Node* __restrict__ nodes, int num_nodes, const ptd::Geom* __restrict__ geoms,
                               f3 o, f3 d, bool valid, int lane, const float* qo_tab, bool tri) {
  const RayInv ri = Ar<EX>::ray_inv(d, o);
  w.best[lane] = kNoHit;
  int nb = 0, ns = 0;  // pending cubes (front of the list) / others (back)
  cnode4* cn = (cnode4*)(uintptr_t)nodes;
  int i = 0;
  // (fetching node i + 1 while node i is tested — the next node whenever some lane passes or a leaf is passed by nobody — was
  // measured: k_primary 1.33 -> 1.39 ms on the 156-primitive scene; the loop is bound by the tests' instructions, not by the loads)
  while (i < num_nodes) {
    const v4f NA = cn[2 * i], NB = cn[2 * i + 1];  // bmin.xyz, bmax.x | bmax.yz, skip, geom
    const bool pass = valid && Ar<EX>::slab(o, ri, NA.x, NA.y, NA.z, NA.w, NB.x, NB.y);
    const unsigned long long m = ballot(pass);
    const int gi = __float_as_int(NB.w);
    if (!m) {
      i = __float_as_int(NB.z);
      continue;
    }
    if (gi >= 0) {  // a leaf some lane passes: type is wave-uniform
      if (nb + ns + 64 > kCandCap) {
        flush_candidates<true, false, EX>(w, nb, ns, lane, o, d, nodes, geoms, qo_tab, tri);
        nb = ns = 0;
      }
      const int rank = rank_in(m), cnt = __popcll(m);
      const uint32_t entry = ((uint32_t)i << 6) | (uint32_t)lane;
      if (__builtin_amdgcn_readfirstlane(geoms[gi].type) == 1) {
        if (pass) w.list[nb + rank] = entry;
        nb += cnt;
      } else {
        if (pass) w.list[kCandCap - ns - cnt + rank] = entry;
        ns += cnt;
      }
    }
    ++i;
  }
  if (nb + ns) flush_candidates<true, false, EX>(w, nb, ns, lane, o, d, nodes, geoms, qo_tab, tri);
}

template <bool TABLES_IN_LDS, bool EX>
__global__ __launch_bounds__(kBlock) void k_intersect(SceneTables sc, ptd::Queues qs, const int32_t* __restrict__ cnt_in,
                                                      ptd::PathBuf paths, ptd::HitBuf hits) {
  extern __shared__ float4 lds_raw[];
  char* lds = reinterpret_cast<char*>(lds_raw);
  // LDS map: [top list][nodes][geoms] (tables, if they fit) then one WaveLds block per wave
  const int nb_top = sc.num_top * (int)sizeof(ptd::TopEntry);
  stage16(lds, EX ? sc.top : sc.top_b, nb_top);  // EX: primary rays, the reference's arithmetic on the reference's boxes; else the build's bounce tables
  const float4* top = reinterpret_cast<const float4*>(lds);
  const ptd::Node* nodes = EX ? sc.nodes : sc.nodes_b;
  const ptd::Geom* geoms = sc.geoms;
  int tbl = nb_top;
  if (TABLES_IN_LDS) {
    const int nb_nodes = sc.num_nodes * (int)sizeof(ptd::Node);
    const int nb_geoms = sc.num_geoms * (int)sizeof(ptd::Geom);
    stage16(lds + nb_top, nodes, nb_nodes);
    stage16(lds + nb_top + nb_nodes, sc.geoms, nb_geoms);
    nodes = reinterpret_cast<const ptd::Node*>(lds + nb_top);
    geoms = reinterpret_cast<const ptd::Geom*>(lds + nb_top + nb_nodes);
    tbl += nb_nodes + nb_geoms;
  }
  __syncthreads();
  const int wib = threadIdx.x >> 6;
  WaveLds w;
  {
    char* base = lds + tbl + wib * kWaveLds;
    w.best = reinterpret_cast<unsigned long long*>(base);
    w.rec = reinterpret_cast<float*>(base + 64 * 8);
    w.list = reinterpret_cast<uint32_t*>(base + 64 * 8 + 7 * 64 * 4);
  }
  const int ntop = sc.num_top;

  const int wave = blockIdx.x * kWavesPerBlock + wib;
  const int lane = lane_id();
  const int q = wave % qs.Q, r = wave / qs.Q, wq = qs.W / qs.Q;
  const int n_q = cnt_in[(size_t)q * qs.cnt_stride];
  const int64_t HS = hits.stride;
  const int64_t qbase = (int64_t)q * qs.cap;
  // Memory operations are kept branch-free so that the compiler can count them (s_waitcnt vmcnt(N)
  // is in-order): every lane of a group loads and stores, lanes past the queue's fill level touch the
  // unused tail of the queue's own region (cap is a multiple of 64, so roundup64(n_q) <= cap) and
  // their results are ignored.  The NEXT group's rays are loaded before the current group is traced
  // (software pipeline): their HBM latency overlaps ~1k instructions of work.
  const int last = qs.cap - 64 + lane;  // clamp for the prefetch beyond the last group
  f3 no, nd;
  path_load_ray(paths, qbase + min(r * 64 + lane, last), no, nd);
  for (int j = r; j * 64 < n_q; j += wq) {
    const int i = j * 64 + lane;
    const bool valid = i < n_q;
    const int64_t at = qbase + i;
    const f3 o = no, d = nd;
    path_load_ray(paths, qbase + min((j + wq) * 64 + lane, last), no, nd);
    trace_group<false, false, EX>(w, top, ntop, nodes, geoms, o, d, valid, lane, sc.cull_margin, sc.top_xor, nullptr, sc.has_triangles != 0);

    const unsigned long long best = w.best[lane];
    const bool hit = (uint32_t)(best >> 32) != 0x7f7fffffu;
    // record layout of the reference after its per-depth memset (pathtrace.cu:562):
    // miss → t = -1 and zeros elsewhere.
    const int leaf = hit ? (int)(uint32_t)best : 0;
    hits.t[at] = hit ? __uint_as_float((uint32_t)(best >> 32)) : -1.0f;
    hits.n[at] = hit ? w.rec[0 * 64 + lane] : 0.f;
    hits.n[HS + at] = hit ? w.rec[1 * 64 + lane] : 0.f;
    hits.n[2 * HS + at] = hit ? w.rec[2 * 64 + lane] : 0.f;
    hits.mat[at] = hit ? geoms[nodes[leaf].geom].material : 0;
    hits.p[at] = hit ? w.rec[3 * 64 + lane] : 0.f;
    hits.p[HS + at] = hit ? w.rec[4 * 64 + lane] : 0.f;
    hits.p[2 * HS + at] = hit ? w.rec[5 * 64 + lane] : 0.f;
  }
}

#include "pt_shade.inc"
// ── candidate ring with carry-over (fused kernels) ────────────────────────────────────────
// In the fused kernels the candidates of consecutive groups share one per-wave FIFO ring: a primitive-
// test chunk is run whenever 64 entries are pending, and what is left at the end of a group's search
// (< 64 entries) waits for the next group's candidates instead of being run as a mostly empty chunk.
// A typical depth->=1 group produces ~55 cube + ~12 sphere candidates — one entry too many for one wave —
// so without carry-over every group paid a full cube chunk plus a sphere chunk at ~15 % lane occupancy.
// Entries carry the lane that owns their ray and the parity of their group; the rays themselves (6 floats), like
// the results (closest-hit keys and records), are kept per lane and double-buffered by parity, and a group is shaded one loop iteration later, after every
// entry appended during its search has been processed (forced partial chunk only if the ring never
// filled up in between).  Order of evaluation still does not matter: the (t, leaf) key minimum is the
// reference's choice.
constexpr int kRing = 128;  // ring entries per wave (power of two; <= 63 pending + <= 64 appended at once)
// SMALL: the primary kernel's ring for LDS-table scenes (every leaf is one of <= 32 top entries, so leaf indices are < 64; every
// ray starts at the camera): 16-bit ring entries, no work-stealing table, and only the rays' DIRECTIONS kept — 5888 B per
// wave, which lets a fifth workgroup of k_primary fit a CU.
// NPAR: 2 = keys / records / rays double-buffered by group parity (k_primary's ring: a group is shaded while the next one is
// searched); 1 = one group at a time (k_paths modes 1 and 2, k_primary on the grid).
template <bool SMALL, int NPAR = 2>
struct Carry {
  using Ent = typename std::conditional<SMALL, uint16_t, uint32_t>::type;
  unsigned long long* best;  // [NPAR][64]
  float* rec;                // [NPAR][6][64]  normal xyz, point xyz
  Ent* ent;                  // [kRing]     (leaf << 7) | (parity << 6) | owner lane
  float* ray;                // [NPAR][kRayPlanes][64]  (origin xyz,) direction xyz of each lane's ray, by group parity
  f3 cam_o;                  // SMALL: the origin of every ray (wave-uniform)
  static constexpr int kRayPlanes = SMALL ? 3 : 6;
  int* slot;                 // [64]        scratch of the work-stealing step (carry_search); not SMALL only
  uint32_t* gix;             // [kRing]     LEAN chunks (grid walk) only: geom index | primitive type << 30 of each ring entry
  const float* qo_tab;       // QO chunks (primary rays) only: the camera position in every geom's object space, [geom][3]
  int head, count;           // wave-uniform
  int appended, processed;   // running totals (wave-uniform)
  int debug;                 // BatchInfo::debug
  const __attribute__((address_space(3))) v4f* lnodes;  // the threaded nodes in LDS (k_paths mode 1 on scenes of a few hundred nodes) when lds_nodes
  bool lds_nodes;
};
template <bool SMALL, int NPAR = 2>
__host__ __device__ constexpr int carry_bytes() {
  return NPAR * 64 * 8 + NPAR * 6 * 64 * 4 + kRing * (SMALL ? 2 : 4) + NPAR * (SMALL ? 3 : 6) * 64 * 4 + (SMALL ? 0 : 64 * 4);
}
template <bool SMALL, int NPAR = 2>
PT_DEV Carry<SMALL, NPAR> carry_init(char* base) {
  Carry<SMALL, NPAR> c;
  c.best = reinterpret_cast<unsigned long long*>(base);
  c.rec = reinterpret_cast<float*>(base + NPAR * 64 * 8);
  c.ray = reinterpret_cast<float*>(base + NPAR * 64 * 8 + NPAR * 6 * 64 * 4);
  constexpr int ray_bytes = NPAR * Carry<SMALL, NPAR>::kRayPlanes * 64 * 4;
  c.ent = reinterpret_cast<typename Carry<SMALL, NPAR>::Ent*>(base + NPAR * 64 * 8 + NPAR * 6 * 64 * 4 + ray_bytes);
  c.slot = reinterpret_cast<int*>(base + NPAR * 64 * 8 + NPAR * 6 * 64 * 4 + ray_bytes + kRing * 4);
  c.cam_o = mk(0.f, 0.f, 0.f);
  c.head = c.count = c.appended = c.processed = 0;
  c.debug = 0;
  c.gix = nullptr;
  c.qo_tab = nullptr;
  c.lnodes = nullptr;
  c.lds_nodes = false;
  return c;
}
// Primitive tests for the first n (<= 64) pending entries; wave-uniform control flow, all lanes active.
// LEAN (the grid walk): the geom index and primitive type come with the ring entry (Carry::gix) instead of through
// nodes[leaf] and the geom record, and the world-space normal is computed for the winner only (finish_normal): 6
// 16-byte reads per candidate instead of 9 (cube) or 11 (sphere) — the L1's access rate is what bounds those kernels.
// QO (the primary kernel's ring): every ray starts at the camera, whose object-space image per geom comes from Carry::qo_tab.
template <bool SMALL, int NPAR, bool EX = false, bool LEAN = false, bool QO = false>
PT_DEV void carry_chunk(Carry<SMALL, NPAR>& c, int n, int lane, const ptd::Node* __restrict__ nodes,
                        const ptd::Geom* __restrict__ geoms) {
  const bool valid = lane < n;
  PT_STAT(6, 1);
  PT_STAT(7, n);
  const int idx = (c.head + lane) & (kRing - 1);
  const uint32_t entry = (uint32_t)c.ent[idx];
  const int src = (int)(entry & 63u);
  const int par = (int)((entry >> 6) & 1u);
  const uint32_t leaf = entry >> 7;
  constexpr int RP = Carry<SMALL, NPAR>::kRayPlanes;
  const float* ray = c.ray + par * RP * 64 + src;
  const f3 ro = SMALL ? c.cam_o : mk(ray[0 * 64], ray[1 * 64], ray[2 * 64]);
  const f3 rd = mk(ray[(RP - 3) * 64], ray[(RP - 2) * 64], ray[(RP - 1) * 64]);
  const uint32_t gw = LEAN ? c.gix[idx] : 0u;
  const int gi = valid ? (LEAN ? (int)(gw & 0x3fffffffu) : nodes[leaf].geom) : 0;
  const ptd::Geom* G = geoms + gi;
  f3 pt = mk(0.f, 0.f, 0.f), nrm = mk(0.f, 0.f, 0.f);
  float t = -1.0f;
  // cube / sphere decided per lane; shared pre and post parts
  if (!(kAblate && (c.debug & 4))) {
    if (LEAN) t = Ar<EX>::template geom_test<-1, false, true>(G, ro, rd, pt, nrm, mk(0.f, 0.f, 0.f), valid ? (int)(gw >> 30) : 0);
    else if (QO) t = Ar<EX>::template geom_test<-1, true>(G, ro, rd, pt, nrm, mk(c.qo_tab[3 * gi], c.qo_tab[3 * gi + 1], c.qo_tab[3 * gi + 2]));
    else t = Ar<EX>::template geom_test<-1, false>(G, ro, rd, pt, nrm, mk(0.f, 0.f, 0.f));
  }
  const uint32_t tb = __float_as_uint(t);
  if (valid && t > 0.f && tb < 0x7f7fffffu) {
    const unsigned long long key = ((unsigned long long)tb << 32) | leaf;
    unsigned long long* slot = &c.best[par * 64 + src];
    atomicMin(slot, key);
    if (*slot == key) {  // this candidate is the ray's best so far: publish its record
      float* r = c.rec + par * 6 * 64 + src;
      r[0 * 64] = nrm.x, r[1 * 64] = nrm.y, r[2 * 64] = nrm.z;
      r[3 * 64] = pt.x, r[4 * 64] = pt.y, r[5 * 64] = pt.z;
    }
  }
  c.head = (c.head + n) & (kRing - 1);
  c.count -= n;
  c.processed += n;
}
// Append the lanes with `pass` (entry: leaf index, group parity, lane that owns the ray); runs a chunk as soon
// as 64 entries are pending.  Wave-uniform control flow.
template <bool SMALL, int NPAR, bool EX = false, bool LEAN = false>
PT_DEV void carry_append(Carry<SMALL, NPAR>& c, bool pass, uint32_t leaf, int par, int owner, int lane,
                         const ptd::Node* __restrict__ nodes, const ptd::Geom* __restrict__ geoms, uint32_t gword = 0u) {
  const unsigned long long m = ballot(pass);
  if (!m) return;
  const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
  if (pass) {
    const int idx = (c.head + c.count + rank) & (kRing - 1);
    c.ent[idx] = (typename Carry<SMALL, NPAR>::Ent)((leaf << 7) | ((uint32_t)par << 6) | (uint32_t)owner);
    if (LEAN) c.gix[idx] = gword;
  }
  const int cnt = __popcll(m);
  c.count += cnt;
  c.appended += cnt;
  if (c.count >= 64) carry_chunk<SMALL, NPAR, EX, LEAN>(c, 64, lane, nodes, geoms);
}
// Candidate search of one group (phase 1 of trace_group) feeding the ring.
// SUB: the scene has subtrees below the top list.  The LDS-table kernels are only used for scenes whose leaves all
// fit the top list (auto_lds_table_limit), so their instantiation drops the subtree scan.
// CAM (primary rays, !SUB only): `top` holds the boxes relative to the camera position (slab_rel), EX selects the
// reference's exact arithmetic, and the chunks take the object-space origin from Carry::qo_tab.
template <bool SUB, int NPAR, bool CAM = false, bool EX = false>
PT_DEV void carry_search(Carry<!SUB, NPAR>& c, const float4* top, int ntop, const ptd::Node* __restrict__ nodes,
                         const ptd::Geom* __restrict__ geoms, f3 o, f3 d, bool valid, int lane, int par, float cull,
                         unsigned long long top_xor) {
  static_assert(CAM == !SUB, "the top-list-only form is the primary kernel's camera ring (Carry<true>: origins not stored); subtree scans take any origin");
  const RayInv ri = Ar<EX>::ray_inv(d, o);
  {
    constexpr int RP = Carry<!SUB, NPAR>::kRayPlanes;
    float* ray = c.ray + par * RP * 64 + lane;  // this group's rays, read back by the primitive-test chunks
    if (RP == 6) ray[0 * 64] = o.x, ray[1 * 64] = o.y, ray[2 * 64] = o.z;
    ray[(RP - 3) * 64] = d.x, ray[(RP - 2) * 64] = d.y, ray[(RP - 1) * 64] = d.z;
  }
  if (!SUB) {
    // Every top entry is a leaf (the LDS-table kernels): first all box tests — one bit per entry in a per-lane mask, boxes
    // fetched one entry ahead into alternating registers (unrolled by two, no copies) — then the appends lane-major: in
    // round j every lane that still has candidates files one, so the ballot / rank / ring arithmetic runs once per round
    // (max candidates of a lane, ~3) instead of once per entry (7 for cornell.txt).  The order of the ring entries changes,
    // the set does not, and the closest-hit key is order-independent.
    // box tests, eight at a time fully unrolled (no loop-carried box registers to rotate); the pass bit is shifted into the mask
    // by ONE v_addc (push_bit): bit (ntop - 1 - e) of the mask = entry e
    uint32_t mask = 0;
    for (int e0 = 0; e0 < ntop; e0 += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + u;
        if (e < ntop) {
          const float4 A = top[2 * e], B = top[2 * e + 1];
          mask = push_bit(mask, CAM ? Ar<EX>::slab_rel(ri, A.x, A.y, A.z, A.w, B.x, B.y) : Ar<EX>::slab(o, ri, A.x, A.y, A.z, A.w, B.x, B.y));
        }
      }
    }
    mask = valid ? mask : 0u;
    const uint32_t tag = ((uint32_t)par << 6) | (uint32_t)lane;
    while (true) {
      const unsigned long long m = ballot(mask != 0u);
      if (!m) break;
      const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
      if (mask != 0u) {
        const int te = ntop - 1 - __builtin_ctz(mask);
        mask &= mask - 1u;
        const uint32_t leaf = __float_as_uint(reinterpret_cast<const float*>(top)[8 * te + 6]);  // TopEntry::idx
        c.ent[(c.head + c.count + rank) & (kRing - 1)] = (typename Carry<!SUB, NPAR>::Ent)((leaf << 7) | tag);
      }
      const int cnt = __popcll(m);
      c.count += cnt;
      c.appended += cnt;
      if (c.count >= 64) carry_chunk<!SUB, NPAR, EX, false, CAM>(c, 64, lane, nodes, geoms);
    }
    return;
  }
  uint32_t pend = 0;  // per lane: top entries that are subtrees and whose box this ray passes
  PT_STAT(0, 1);
  PT_STAT(13, __popcll(ballot(valid)));
  float4 A = top[0], B = top[1];
  for (int e = 0; e < ntop; ++e) {
    const float4 TA = A, TB = B;  // bmin.xyz, bmax.x | bmax.yz, idx, link
    if (e + 1 < ntop) A = top[2 * e + 2], B = top[2 * e + 3];
    const int t_idx = __builtin_amdgcn_readfirstlane(__float_as_int(TB.z));
    const int t_link = __builtin_amdgcn_readfirstlane(__float_as_int(TB.w));
    const bool pass = valid && slab(o, ri, TA.x, TA.y, TA.z, TA.w, TB.x, TB.y);
    PT_STAT(8, __popcll(ballot(pass)));
    // (the closer-hit cull on top entries that are leaves — scenes of a few dozen primitives, where all of them are — was measured:
    // 32 primitives 9.9 -> 9.2 k Msamples/s; the list is not in near-first order and every entry pays the LDS read of the best hit)
    if (t_link < 0) carry_append(c, pass, (uint32_t)t_idx, par, lane, lane, nodes, geoms);
    else if (pass) pend |= 1u << e;
  }
  // Subtrees below the cut (large scenes only): per-lane stackless scans, nearest subtree first, with work stealing
  // (scan_next / steal_step); candidates are filed under the lane that owns the ray, so nothing downstream changes.
  if (SUB && ballot(pend != 0)) {
    Walker wk{0, 0, lane, o, ri};
    const uint32_t xm = octant_mask(ri, top_xor);
    pend = permute_xor(pend, xm);
    while (true) {
      if (wk.cur >= wk.end && pend) {  // own subtrees first, nearest first
        const int e = __builtin_ctz(pend) ^ (int)xm;
        pend &= pend - 1;
        const float4 TB = top[2 * e + 1];
        wk.cur = __float_as_int(TB.z) + 1;  // the subtree root's box is the top entry's box: already passed
        wk.end = __float_as_int(TB.w);
      }
      const bool idle = wk.cur >= wk.end;
      const unsigned long long I = ballot(idle);
      if (I == ~0ull) break;
      if (__popcll(I) >= kStealMin) {
        PT_STAT(3, 1);
        steal_step(wk, pend, xm, idle, I, c.slot, top, lane);
      }
      const bool act = wk.cur < wk.end;
      PT_STAT(1, 1);
      PT_STAT(2, __popcll(ballot(act)));
      const float bt = __uint_as_float(reinterpret_cast<const uint32_t*>(c.best)[2 * (par * 64 + wk.own) + 1]) + cull;
      bool cand;
      int at_n, aux;
      if (c.lds_nodes) scan_step(c.lnodes, wk.o, wk.ri, act, wk.cur, bt, cand, at_n, aux);
      else scan_step(reinterpret_cast<const v4f*>(nodes), wk.o, wk.ri, act, wk.cur, bt, cand, at_n, aux);
      carry_append(c, cand, (uint32_t)at_n, par, wk.own, lane, nodes, geoms);
    }
  }
}
// Make sure everything appended up to `mark` has been tested (only runs a partial chunk when the ring
// did not fill up since).
template <bool SMALL, int NPAR, bool EX = false, bool QO = false>
PT_DEV void carry_drain_to(Carry<SMALL, NPAR>& c, int mark, int lane, const ptd::Node* __restrict__ nodes,
                           const ptd::Geom* __restrict__ geoms) {
  while (c.processed - mark < 0) carry_chunk<SMALL, NPAR, EX, false, QO>(c, min(64, c.count), lane, nodes, geoms);
}

#include "pt_grid.inc"
// ── depth 0 fused: generateRayFromCamera + computeIntersections + shadeAndExtendRays ────────
// Primary rays are a pure function of the sample id, so depth 0 needs no path state in memory at
// all: the ray is built in registers, traced with the same wave-cooperative search, shaded, and
// only its outcome is written — the survivor's new ray into the depth-1 queues or the retired
// colour.  That removes the generate launch and ~190 B/sample of HBM round trips (40 B ray state
// written + 24 B read, 32 B hit record written + read, 28 B path state re-read) at the one depth
// where every sample is alive.  Also writes the per-queue sample counts of depth 0 (statistics).
// GRID: large scenes with a uniform grid over the leaf boxes (SceneTables::use_grid): the primary rays walk it like the
// bounce rays do (grid_search) instead of testing the top list and scanning subtrees; exact arithmetic as in every depth-0 path.
// RING (the LDS-table scenes, e.g. cornell.txt): the candidates go through the bounce kernel's ring with carry-over
// (carry_search / carry_chunk), so that primitive tests only ever run as full 64-entry chunks — a group of primary rays
// files ~85 candidates, which the per-group form (trace_group) ran as one full chunk plus one a third full — and the
// appends run lane-major (two-phase search).  A group is shaded one loop iteration after its search, like in k_bounce.
// Unlike the ring form tried in round 2 it keeps the camera-relative boxes and the per-geom object-space camera position.
#ifndef PT_PRIMARY_RING
#define PT_PRIMARY_RING 1
#endif
#ifndef PT_PRIMARY_PACKET
#define PT_PRIMARY_PACKET 1  // global-table scenes: one wave-uniform scan of the threaded tree per group (trace_group_packet) instead of top list + per-lane subtree scans
#endif
template <bool TABLES_IN_LDS, bool GRID>
constexpr bool primary_ring() { return PT_PRIMARY_RING != 0 && TABLES_IN_LDS && !GRID; }
template <bool TABLES_IN_LDS, bool GRID = false>
__global__ __launch_bounds__(kBlock, kPrimaryWaves) void k_primary(SceneTables sc, ptd::Camera cam, BatchInfo b, ptd::Queues qs,
                                                    int32_t* __restrict__ cnt0, int32_t* __restrict__ cnt_out,
                                                    ptd::PathBuf out, ptd::RetireBuf ret) {
  extern __shared__ float4 lds_raw[];
  char* lds = reinterpret_cast<char*>(lds_raw);
  static_assert(!(GRID && TABLES_IN_LDS), "the grid walk reads the tables from memory");
  const int nb_top = GRID ? 0 : sc.num_top * (int)sizeof(ptd::TopEntry);
  const int nb_mats = (sc.num_mats * (int)sizeof(ptd::Mat) + 15) & ~15;
  stage16(lds, sc.top, nb_top);
  stage16(lds + nb_top, sc.mats, nb_mats);
  const ptd::Mat* mats = reinterpret_cast<const ptd::Mat*>(lds + nb_top);
  const ptd::Node* nodes = sc.nodes;
  const ptd::Geom* geoms = sc.geoms;
  int tbl = nb_top + nb_mats;
  if (TABLES_IN_LDS) {
    const int nb_nodes = sc.num_nodes * (int)sizeof(ptd::Node);
    const int nb_geoms = sc.num_geoms * (int)sizeof(ptd::Geom);
    stage16(lds + tbl, sc.nodes, nb_nodes);
    stage16(lds + tbl + nb_nodes, sc.geoms, nb_geoms);
    nodes = reinterpret_cast<const ptd::Node*>(lds + tbl);
    geoms = reinterpret_cast<const ptd::Geom*>(lds + tbl + nb_nodes);
    tbl += nb_nodes + nb_geoms;
  }
  constexpr bool RING = primary_ring<TABLES_IN_LDS, GRID>();
  constexpr int kWaveBytes = GRID ? grid_wave_bytes<kD0>() : (RING ? carry_bytes<true, 2>() : kWaveLds);
  uint32_t* ihash = reinterpret_cast<uint32_t*>(lds + tbl + kWavesPerBlock * kWaveBytes);  // after the per-wave blocks
  iter_hash_fill(ihash, sc, b, 0);
  // camera-relative copies for the primary rays: top-list boxes minus the camera position and (tables in LDS only)
  // the camera position in each geom's object space — the same float operations the per-ray code would execute
  float4* cam_top = reinterpret_cast<float4*>(ihash + iter_hash_entries(sc));
  float* cam_qo = reinterpret_cast<float*>(cam_top + 2 * sc.num_top);
  {
    const f3 cp = mk(cam.pos[0], cam.pos[1], cam.pos[2]);
    for (int e = threadIdx.x; !GRID && e < sc.num_top; e += blockDim.x) {
      const ptd::TopEntry t = sc.top[e];
      cam_top[2 * e] = make_float4(t.bmin[0] - cp.x, t.bmin[1] - cp.y, t.bmin[2] - cp.z, t.bmax[0] - cp.x);
      cam_top[2 * e + 1] = make_float4(t.bmax[1] - cp.y, t.bmax[2] - cp.z, __int_as_float(t.idx), __int_as_float(t.link));
    }
    for (int gi = threadIdx.x; TABLES_IN_LDS && gi < sc.num_geoms; gi += blockDim.x) {
      const f3 q = Ar<kD0>::xf_point(sc.geoms[gi].inv, cp);
      cam_qo[3 * gi] = q.x, cam_qo[3 * gi + 1] = q.y, cam_qo[3 * gi + 2] = q.z;
    }
  }
  __syncthreads();
  const int wib = threadIdx.x >> 6;
  WaveLds w;
  Carry<false, 1> cy = carry_init<false, 1>(lds + tbl + wib * kWaveBytes);  // GRID: the bounce kernel's rings
  CellRing cr{reinterpret_cast<uint32_t*>(lds + tbl + wib * kWaveBytes + carry_bytes<false, 1>()), 0, 0, nullptr};
  cy.debug = b.debug;
  if (GRID) {
    cy.gix = cr.ent + kCellRing;
    cr.rinv = reinterpret_cast<float*>(cr.ent + kCellRing + kRing);
    w.best = cy.best;
    w.rec = cy.rec;
    w.list = nullptr;
  } else {
    char* base = lds + tbl + wib * kWaveBytes;
    w.best = reinterpret_cast<unsigned long long*>(base);
    w.rec = reinterpret_cast<float*>(base + 64 * 8);
    w.list = reinterpret_cast<uint32_t*>(base + 64 * 8 + 7 * 64 * 4);
  }
  const int ntop = sc.num_top;
  const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + wib);  // (the compiler cannot see that threadIdx.x >> 6 is wave-uniform)
  const int lane = lane_id();
  // A strand = what one wave of one queue traces at depth 0: in iteration k the queue's chunks of residue (r + k) mod wq.  Wave w
  // runs strand w.  With a small tile (a rank's share of a frame: a queue owns a dozen chunks, some of them beside the scene,
  // which cost a bounds test) the strands differ by +-20 %: they are then cut into `pieces` runs of iterations — every
  // (queue, iteration, residue) still has ONE owner, which is all the sub-lists ask for — piece 0 stays with wave w and the
  // others go to whoever is free, in the order of a counter (zeroed by k_count_stats).
  const int kp = b.primary_pieces > 1 && qs.deal != nullptr && !b.flat ? (b.K + b.primary_pieces - 1) / b.primary_pieces : b.K;
  const int pieces = (b.K + kp - 1) / kp;
  for (int strand = wave; strand < qs.W * pieces;) {
    const int piece = strand / qs.W, sw = strand - piece * qs.W;
    const int k0 = piece * kp, k1 = min(b.K, k0 + kp);
    const int q = sw % qs.Q, r = sw / qs.Q, wq = qs.W / qs.Q;
    const Retire rt = retire_of(ret, q);
    // The wave's samples: in iteration k the chunks jj = rho, rho + wq, ... of the queue's my_nq (chunk jj = tile chunk q + jj * Q),
    // rho = (r + k) mod wq — its own sub-list and sub-region of (q, k) (pt_device.h RetireBuf): positions come from the two
    // counters below, nothing is reserved with atomics and every store is issued where its group is shaded.
    const QueueShare sh = queue_share(b, qs, q);
    if (piece == 0 && r == 0 && lane == 0) cnt0[(size_t)q * qs.cnt_stride] = b.K * sh.my_pixels;
    const int64_t qbase = (int64_t)q * qs.cap;
    const float inv_w = 1.0f / (float)cam.res_x;
    const f3 o = mk(cam.pos[0], cam.pos[1], cam.pos[2]);
    int32_t* counter = &cnt_out[(size_t)q * qs.cnt_stride];  // flat form: the queue's ONE depth-1 list
    const int quo = sh.my_nq / wq, rem = sh.my_nq % wq;
    int ck = k0 - 1, crho = (r + k0 + wq - 1) % wq, nl = 0, nd = 0;  // iteration the counters belong to, its residue (r + ck) mod wq; survivors / retirees of the wave in it so far
    auto publish = [&]() {  // the finished iteration's counts (also when the wave had no chunk in it: zeros)
      if (ck >= k0 && lane == 0) rt.sub[ck * rt.wq0 + crho] = ((unsigned long long)(uint32_t)nd << 32) | (uint32_t)nl;
    };
    auto next_iteration = [&]() {  // publish, then on to ck + 1
      publish();
      nl = nd = 0;
      ++ck;
      crho = crho + 1 == wq ? 0 : crho + 1;
    };
    // shading + retirement + compaction of one group of primary rays from its resolved hit key / record
    auto shade_group = [&](unsigned long long best, const float* rec, bool valid, int k, int pl, int slot, uint32_t phash, f3 d) {
      const bool hit = (uint32_t)(best >> 32) != 0x7f7fffffu;
      ShadeIO s;
      s.o = o;
      s.d = d;
      s.c = mk(1.0f, 1.0f, 1.0f);
      s.alive = false;
      Bounce bo;
      bo.kind = 0;
      f3 hn = mk(0.f, 0.f, 0.f), hp = mk(0.f, 0.f, 0.f);
      if (valid) {
        float ht = -1.0f;
        int hmat = 0;
        if (hit) {
          ht = __uint_as_float((uint32_t)(best >> 32));
          const ptd::Geom* G = geoms + nodes[(uint32_t)best].geom;
          hmat = G->material;
          hn = mk(rec[0 * 64], rec[1 * 64], rec[2 * 64]);
          hp = mk(rec[3 * 64], rec[4 * 64], rec[5 * 64]);
          if (GRID) hn = Ar<kD0>::finish_normal(G, hn);  // the grid's chunks leave the normal to the winner (carry_chunk, LEAN)
        }
        bo = shade_decide(mats, b.trace_depth, 0, iter_hash_of(ihash, sc, b, 0, k) ^ phash, ht, hmat, s);
      }
      const bool alive = valid && s.alive, dead = valid && !s.alive;
      if (alive) shade_bounce(bo, hn, hp, s);
      const unsigned long long live = ballot(alive);
      const PathTag tag{slot, phash, k};
      if (b.flat) {  // unfused consumers: one dense list per queue behind an atomic, records appended one by one (test / A-B form)
        int base = 0;
        if (live && lane == 0) base = atomicAdd(counter, (int)__popcll(live));
        base = __builtin_amdgcn_readfirstlane(base);
        if (alive) path_store(out, qbase + base + rank_in(live), s.o, s.d, s.c, tag);
        retire_append(rt, dead, k, pl, s.c);
        return;
      }
      while (ck < k) next_iteration();  // first group of a new iteration: the previous one's counts are final (iterations without a chunk publish zeros on the way)
      const int sub0 = k * rt.seg_cap + sub_offset(quo, rem, crho) * 64;  // first slot of sub-list / sub-region (q, k, rho)
      const unsigned long long deadm = ballot(dead);
      if (alive) path_store(out, qbase + sub0 + nl + rank_in(live), s.o, s.d, s.c, tag);
      if (dead) rt.rec[sub0 + nd + rank_in(deadm)] = ptd::Word4{s.c.x, s.c.y, s.c.z, __int_as_float(pl)};
      nl += (int)__popcll(live), nd += (int)__popcll(deadm);
    };
    // a group between its search and its shading (RING)
    struct {
      f3 d;
      int k, pl, slot;
      uint32_t phash;
      bool valid;
      int par, mark;
      bool any;
    } pp;
    pp.any = false;
    Carry<true, 2> rc = carry_init<true, 2>(lds + tbl + wib * kWaveBytes);  // RING only (the same bytes as `w` otherwise)
    rc.debug = b.debug;
    rc.qo_tab = cam_qo;
    rc.cam_o = o;
    int it = 0;
    for (int k = k0, rho = (r + k0) % wq; k < k1; ++k, rho = rho + 1 == wq ? 0 : rho + 1) {
      for (int jj = rho; jj < sh.my_nq; jj += wq, ++it) {
        const int pl_raw = (q + jj * qs.Q) * 64 + lane;
        const bool valid = pl_raw < b.N;
        const int pl = valid ? pl_raw : b.N - 1;  // tile pixel
        const int slot = make_slot(b, k, pl);
        const int p = global_pixel(b, pl);  // global pixel index
        const uint32_t phash = utilhash((uint32_t)p);
        float jx = 0.f, jy = 0.f;
        if (b.aa_jitter) aa_jitter(b.iter_first + k, p, jx, jy);
        const f3 d = Ar<kD0>::camera_dir(cam, inv_w, p, b.aa_jitter != 0, jx, jy);
        // Primary rays come in bundles of 64 neighbouring pixels and half of the 16:9 frame looks past the scene:
        // one test against the bounds of the whole tree per lane, and if no lane passes (a parent box rejects
        // whatever its children would, the slab arithmetic being monotone) the 7 leaf-box tests are skipped.
        const bool near_scene = ballot(valid && Ar<kD0>::slab(o, Ar<kD0>::ray_inv(d, o), sc.root_min[0], sc.root_min[1], sc.root_min[2],
                                                       sc.root_max[0], sc.root_max[1], sc.root_max[2])) != 0;
        if constexpr (RING) {
          const int par = it & 1;
          rc.best[par * 64 + lane] = kNoHit;
          if (near_scene) carry_search<false, 2, true, kD0>(rc, cam_top, ntop, nodes, geoms, o, d, valid, lane, par, sc.cull_margin, sc.top_xor);
          if (pp.any) carry_drain_to<true, 2, kD0, true>(rc, pp.mark, lane, nodes, geoms);  // the previous group's candidates are now all resolved
          if (pp.any) shade_group(rc.best[pp.par * 64 + lane], rc.rec + pp.par * 6 * 64 + lane, pp.valid, pp.k, pp.pl, pp.slot, pp.phash, pp.d);
          pp.d = d, pp.k = k, pp.pl = pl, pp.slot = slot, pp.phash = phash, pp.valid = valid, pp.par = par, pp.mark = rc.appended, pp.any = true;
        } else {
          if (GRID) {
            w.best[lane] = kNoHit;
            if (near_scene) {
              grid_search<1, kD0>(cy, cr, sc, nodes, geoms, o, d, valid, lane, 0);
              while (cy.count > 0) carry_chunk<false, 1, kD0, true>(cy, min(64, cy.count), lane, nodes, geoms);
            }
          } else if (near_scene) {
            if constexpr (!TABLES_IN_LDS && PT_PRIMARY_PACKET != 0) trace_group_packet<kD0>(w, nodes, sc.num_nodes, geoms, o, d, valid, lane, cam_qo, sc.has_triangles != 0);
            else trace_group<true, TABLES_IN_LDS, kD0>(w, cam_top, ntop, nodes, geoms, o, d, valid, lane, sc.cull_margin, sc.top_xor, cam_qo, sc.has_triangles != 0);
          } else {
            w.best[lane] = kNoHit;
          }
          shade_group(w.best[lane], w.rec + lane, valid, k, pl, slot, phash, d);
        }
      }
    }
    if constexpr (RING) {
      if (pp.any) {
        carry_drain_to<true, 2, kD0, true>(rc, pp.mark, lane, nodes, geoms);
        shade_group(rc.best[pp.par * 64 + lane], rc.rec + pp.par * 6 * 64 + lane, pp.valid, pp.k, pp.pl, pp.slot, pp.phash, pp.d);
      }
    }
    if (!b.flat)
      while (ck < k1) next_iteration();  // the last iteration's counts, and zeros for trailing iterations without a chunk
    if (pieces == 1) break;
    int nx = 0;
    if (lane == 0) nx = atomicAdd(&qs.deal[2 * qs.Q + 1], 1);
    strand = qs.W + __builtin_amdgcn_readfirstlane(nx);
  }
}

// ── ALL depths >= 1 in one launch: persistent lanes (k_paths) ─────────────────────────────────────────────────────────────────
// After depth 0 a path never goes back to HBM.  A wave has ONE set of 64 lanes; a lane carries its ray, its throughput, its
// sample id and ITS OWN depth.  When a path dies the lane writes the 16-byte retirement record and takes the next depth-1 ray
// of the wave's slice of its queue (k_primary's output); survivors stay where they are, at depth + 1.  The search, the
// candidate ring and the primitive-test chunks are per ray (candidates carry their owner lane), the RNG is keyed per lane by
// (iteration, pixel, depth), the retirement records by iteration: the results cannot differ from the per-depth kernels'.
// Traffic per PATH: its 40-byte depth-1 record read once, its 16-byte retirement record written once.
//   * per-LANE resolution: primitive tests only ever run as full 64-entry chunks, so a search leaves < 64 candidates in the
//     ring.  A lane is shaded as soon as the ring has processed the last candidate it filed (`mark`); the few lanes whose last
//     candidate is still pending sit out this round (~3 of 64 on cornell.txt) and are shaded in the next one, after the new
//     rays' candidates have pushed theirs through a full chunk.  (Rounds 1-3 kept two groups per wave for this and shaded a
//     group one iteration after its search: twice the lane state and LDS, four resident workgroups per CU instead of six.)
//   * refill: every lane owns a 40-byte LDS slot holding the NEXT path record it will take; the slot is filled by the
//     gfx950 LDS-direct loads (global_load_lds_dwordx4 / _dword: memory -> LDS without passing through VGPRs, LDS address =
//     M0 + lane * size), issued when the lane takes the previous one, i.e. at least a whole iteration before it is read.
//     The loads are inline assembly: the compiler's handling of the builtin either waits for the transfer at the very next
//     LDS access (it cannot tell the slot from the rest of the dynamic LDS) or does not compile under a divergent branch,
//     so the transfer, its s_waitcnt and the slot reads are all spelled out here.
//   * every vector-memory operation outstanding at the refill's s_waitcnt vmcnt(0) is a whole iteration old: the retirement
//     store of a lane that died is issued AFTER the slot reads of the next refill (the dead lane's colour and sample id
//     stay in its registers until then).
#ifndef PT_PATHS_WAVES
#define PT_PATHS_WAVES 6
#endif
#ifndef PT_PATHS_MIN_READY
#define PT_PATHS_MIN_READY 32  // fewer resolved lanes than this and candidates pending: run the partial chunk instead of shading a thin group
#endif
constexpr int kPathsWaves = PT_PATHS_WAVES, kPathsMinReady = PT_PATHS_MIN_READY;
constexpr int kSlotBytes = 64 * 16 + 64 * 16 + 64 * 4 + 64 * 4 + 64 * 4;  // planes 0, 1 (16 B per lane), colour.z, sample id, record slot
PT_DEV uint32_t lds_offset(const void* p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p; }
// memory -> LDS without passing through VGPRs: path record i of a queue (b0 / b1 / b2 = the queue's first record in planes 0,
// 1, 2; wave-uniform, so they are scalar bases and a lane supplies 32-bit byte offsets only) of every active lane to lds_base
// (wave-uniform) + {0, 1024} + lane * 16 (planes 0 and 1) and + {2048, 2304} + lane * 4 (colour.z, sample id).  The
// instruction's immediate offset moves BOTH addresses, hence M0 + 0xfc for the last transfer.  M0 (the transfers' LDS base)
// belongs to the compiler: saved and restored.
template <typename T>
PT_DEV const T* uniform_ptr(const T* p) {  // a pointer the compiler cannot prove wave-uniform (derived from threadIdx.x >> 6), into scalar registers
  const uint64_t v = (uint64_t)(uintptr_t)p;
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
  return reinterpret_cast<const T*>((uintptr_t)(((uint64_t)hi << 32) | lo));
}
PT_DEV void fetch_record_to_lds(const void* b0, const void* b1, const void* b2, int i, uint32_t lds_base) {
  uint32_t keep;
  const uint32_t off16 = (uint32_t)i << 4, off8 = (uint32_t)i << 3;
  asm volatile(
      "s_mov_b32 %[keep], m0\n\t"
      "s_mov_b32 m0, %[base]\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %[o16], %[b0]\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %[o16], %[b1]\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dword %[o8], %[b2]\n\t"
      "s_add_u32 m0, m0, 0xfc\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dword %[o8], %[b2] offset:4\n\t"
      "s_mov_b32 m0, %[keep]"
      : [keep] "=&s"(keep)
      : [base] "s"(lds_base), [b0] "s"(b0), [b1] "s"(b1), [b2] "s"(b2), [o16] "v"(off16), [o8] "v"(off8)
      : "memory", "scc");
}
// Per-wave LDS of k_paths: closest-hit keys, winner records, the candidate ring (16-bit entries: top entry << 6 | owner lane;
// the chunk looks leaf and geom index up in tword[top entry] = leaf | geom << 8, so it does not go through nodes[leaf]) and the
// running totals.  The rays are NOT kept in LDS: the lanes are persistent, a chunk fetches a candidate's ray from its owner's
// registers (ds_bpermute).
struct Lanes {
  unsigned long long* best;  // [64]
  float* rec;                // [6][64]  normal xyz, point xyz
  uint16_t* ent;             // [kRing]
  int head, count;           // wave-uniform
  int appended, processed;   // running totals (wave-uniform)
};
constexpr int kLanesBytes = 64 * 8 + 6 * 64 * 4 + kRing * 2;
// Primitive tests for the first n (<= 64) pending entries; wave-uniform control flow, all 64 lanes active.
PT_DEV void paths_chunk(Lanes& c, int n, int lane, f3 o, f3 d, const uint32_t* tword, const ptd::Geom* __restrict__ geoms) {
  const bool valid = lane < n;
  const uint32_t entry = c.ent[(c.head + lane) & (kRing - 1)];
  const int src = (int)(entry & 63u);
  const uint32_t tw = tword[valid ? entry >> 6 : 0u];
  const uint32_t leaf = tw & 255u;
  const f3 ro = mk(bperm(src, o.x), bperm(src, o.y), bperm(src, o.z));
  const f3 rd = mk(bperm(src, d.x), bperm(src, d.y), bperm(src, d.z));
  const ptd::Geom* G = geoms + (valid ? (int)(tw >> 8) : 0);
  f3 pt = mk(0.f, 0.f, 0.f), nrm = mk(0.f, 0.f, 0.f);
  const float t = geom_test<-1, false>(G, ro, rd, pt, nrm, mk(0.f, 0.f, 0.f));
  const uint32_t tb = __float_as_uint(t);
  if (valid && t > 0.f && tb < 0x7f7fffffu) {
    const unsigned long long key = ((unsigned long long)tb << 32) | leaf;
    unsigned long long* slot = &c.best[src];
    atomicMin(slot, key);
    if (*slot == key) {  // this candidate is the ray's best so far: publish its record
      float* r = c.rec + src;
      r[0 * 64] = nrm.x, r[1 * 64] = nrm.y, r[2 * 64] = nrm.z;
      r[3 * 64] = pt.x, r[4 * 64] = pt.y, r[5 * 64] = pt.z;
    }
  }
  c.head = (c.head + n) & (kRing - 1);
  c.count -= n;
  c.processed += n;
}
// Candidate search of the fresh lanes of a persistent group (LDS-table scenes: every top entry is a leaf) — carry_search's
// two-phase form with the per-lane resolution mark: `mark` = ring entries appended up to and including the lane's last one.
// tword[e] = leaf | geom << 8 of top entry e (read by the chunks).
#ifndef PT_TOP_SCALAR
#define PT_TOP_SCALAR (PT_ARITH == 2)  // mode 0, fast build: the top list's boxes come through scalar loads (constant address space: s_load into SGPRs, the scalar cache) instead of
                                       // broadcast LDS reads into VGPRs: 73 -> 67 VGPRs; in-box fast +0.8 %, exact -1.4 % (its longer tests already cover the LDS latency)
#endif
typedef __attribute__((address_space(4))) const v4f cfloat4;  // wave-uniform indices into it become scalar loads (a builtin vector: HIP's float4 class cannot be copied out of another address space)
template <typename TOP>
PT_DEV void paths_search(Lanes& c, TOP* top, const uint32_t* tword, int ntop, const ptd::Geom* __restrict__ geoms, f3 o, f3 d,
                         bool fresh, int lane, int& mark) {
  const RayInv ri = ray_inv(d, o);
  if (fresh) {
    c.best[lane] = kNoHit;
    mark = c.processed;  // no candidate: resolved at once
  }
  // box tests, eight at a time fully unrolled (no loop-carried box registers to rotate); bit (ntop - 1 - e) of the mask = entry e
  uint32_t mask = 0;
  {
    // (fetching four boxes together before testing them — one wait for four scalar loads instead of one per box — was measured in
    // the fast build: k_paths 1292 -> 1315 us, 32 more SGPRs in flight and 19 spilled)
    for (int e0 = 0; e0 < ntop; e0 += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + u;
        if (e < ntop) {
          const auto A = top[2 * e], B = top[2 * e + 1];
          mask = push_bit(mask, slab(o, ri, A.x, A.y, A.z, A.w, B.x, B.y));
        }
      }
    }
  }
  mask = fresh ? mask : 0u;
  while (true) {
    const unsigned long long m = ballot(mask != 0u);
    if (!m) break;
    const int rank = rank_in(m);
    if (mask != 0u) {
      const int te = ntop - 1 - __builtin_ctz(mask);
      mask &= mask - 1u;
      c.ent[(c.head + c.count + rank) & (kRing - 1)] = (uint16_t)((te << 6) | lane);
      if (mask == 0u) mark = c.appended + rank + 1;
    }
    const int cnt = __popcll(m);
    c.count += cnt;
    c.appended += cnt;
    if (c.count >= 64) paths_chunk(c, 64, lane, o, d, tword, geoms);
  }
}
// MODE 0: scene tables in LDS, every leaf a top entry (cornell.txt: the form described above).
// MODE 1: tables in memory, top list + per-lane subtree scans with work stealing (carry_search<true>);
// MODE 2: tables in memory, uniform grid walk (grid_search).  Modes 1 and 2 run every pending candidate before they shade
//         (a search of theirs files several chunks' worth, and stolen work files candidates under other lanes' names, so a
//         per-lane mark would need cross-lane bookkeeping): all live lanes are searched and shaded in every round.
//         The grid walk's rings leave no LDS for refill slots (three workgroups per CU with them: measured 17-22 % slower than
//         the per-depth kernel it replaces), and its rounds take tens of microseconds: the next record of a lane waits in
//         REGISTERS there, loaded a round ahead by ordinary loads (waves per SIMD: see PT_PATHS_GRID_WAVES; prefetching only origin
//         and direction and loading the rest on demand: slower).
#ifndef PT_PATHS_SCAN_WAVES
#define PT_PATHS_SCAN_WAVES 5
#endif
#ifndef PT_PATHS_GRID_WAVES  // 95-96 VGPRs without scratch since the wave index is scalar; in-box 4 and 5 are level for the fast build, 5 wins for exact
#define PT_PATHS_GRID_WAVES 5
#endif
#ifndef PT_SLOTS_MODES
#define PT_SLOTS_MODES 0  // experiment: 0 = only mode 0 keeps refill slots in LDS
#endif
template <int MODE>
constexpr bool paths_slots_in_lds() { return MODE == 0 || (MODE == 1 && PT_SLOTS_MODES == 1); }
template <int MODE>
constexpr int paths_extra_bytes() { return (paths_slots_in_lds<MODE>() ? kSlotBytes : 0) + 512; }  // refill slots + 64 counters: paths retired per depth + 64: record slots per sub-list
template <int MODE>
constexpr int paths_wave_bytes() {
  return (MODE == 0 ? kLanesBytes : MODE == 1 ? carry_bytes<false, 1>() : grid_wave_bytes<false>()) + paths_extra_bytes<MODE>();
}
template <int MODE>
__global__ __launch_bounds__(kBlock, MODE == 0 ? kPathsWaves : MODE == 1 ? PT_PATHS_SCAN_WAVES : PT_PATHS_GRID_WAVES) void k_paths(SceneTables sc, BatchInfo b, ptd::Queues qs, int32_t* __restrict__ cnt /* [depth][Q] rows */,
                                                               ptd::PathBuf in, ptd::RetireBuf ret) {
  extern __shared__ float4 lds_raw[];
  char* lds = reinterpret_cast<char*>(lds_raw);
  const int nb_top = MODE == 2 ? 0 : sc.num_top * (int)sizeof(ptd::TopEntry);  // the grid walk replaces top list and subtrees
  const int nb_mats = (sc.num_mats * (int)sizeof(ptd::Mat) + 15) & ~15;
  stage16(lds, sc.top_b, nb_top);  // the bounce kernels' box tables (SceneTables::*_b)
  stage16(lds + nb_top, sc.mats, nb_mats);
  const float4* top = reinterpret_cast<const float4*>(lds);
  const ptd::Mat* mats = reinterpret_cast<const ptd::Mat*>(lds + nb_top);
  int tbl = nb_top + nb_mats;
  const ptd::Node* nodes = sc.nodes_b;
  const ptd::Geom* geoms = sc.geoms;
  if (MODE == 0) {  // the geometry records; of the nodes only tword / lmat below are needed (every leaf is a top entry)
    const int nb_geoms = sc.num_geoms * (int)sizeof(ptd::Geom);
    stage16(lds + tbl, sc.geoms, nb_geoms);
    geoms = reinterpret_cast<const ptd::Geom*>(lds + tbl);
    tbl += nb_geoms;
  }
  const char* lnodes = nullptr;
  if (MODE == 1 && sc.scan_nodes_lds > 0) {
    const int nb_nodes = sc.num_nodes * (int)sizeof(ptd::Node);
    stage16(lds + tbl, sc.nodes_b, nb_nodes);
    lnodes = lds + tbl;
    tbl += nb_nodes;
  }
  constexpr int wave_bytes = paths_wave_bytes<MODE>();
  constexpr int core_bytes = wave_bytes - paths_extra_bytes<MODE>();
  const int he = iter_hash_entries(sc);
  uint32_t* tword = reinterpret_cast<uint32_t*>(lds + tbl + kWavesPerBlock * wave_bytes);  // MODE 0: [kMaxTop] leaf | geom << 8 per top entry
  int* lmat = reinterpret_cast<int*>(tword + (MODE == 0 ? kMaxTop : 0));                    // MODE 0: [64] material of the leaf at threaded node index i
  uint32_t* ihash = reinterpret_cast<uint32_t*>(lmat + (MODE == 0 ? 64 : 0));               // rows of the depths 1 .. trace_depth - 1
  for (int d = 1; d < b.trace_depth; ++d) iter_hash_fill(ihash + (d - 1) * he, sc, b, d);
  for (int e = threadIdx.x; MODE == 0 && e < sc.num_top; e += blockDim.x) {
    const int leaf = sc.top[e].idx, gi = sc.nodes[leaf].geom;
    tword[e] = (uint32_t)leaf | ((uint32_t)gi << 8);
    lmat[leaf & 63] = sc.geoms[gi].material;
  }
  __syncthreads();
  const int wib = threadIdx.x >> 6;
  char* wbase = lds + tbl + wib * wave_bytes;
  Lanes cy;  // MODE 0
  cy.best = reinterpret_cast<unsigned long long*>(wbase);
  cy.rec = reinterpret_cast<float*>(wbase + 64 * 8);
  cy.ent = reinterpret_cast<uint16_t*>(wbase + 64 * 8 + 6 * 64 * 4);
  cy.head = cy.count = cy.appended = cy.processed = 0;
  Carry<false, 1> cb = carry_init<false, 1>(wbase);  // MODES 1, 2 (the same bytes)
  CellRing cr{reinterpret_cast<uint32_t*>(wbase + carry_bytes<false, 1>()), 0, 0, nullptr};
  if (MODE == 2) cb.gix = cr.ent + kCellRing, cr.rinv = reinterpret_cast<float*>(cr.ent + kCellRing + kRing);
  cb.debug = b.debug;
  cb.lnodes = (const __attribute__((address_space(3))) v4f*)(lnodes ? lnodes : lds), cb.lds_nodes = lnodes != nullptr;
  constexpr bool SLOTS = paths_slots_in_lds<MODE>();
  char* slots = wbase + core_bytes;  // SLOTS: [64] x 16 B, [64] x 16 B, [64] x 4 B, [64] x 4 B
  int* died = reinterpret_cast<int*>(slots + (SLOTS ? kSlotBytes : 0));  // [64]: paths of this wave retired AT depth d (statistics; PT_MAX_DEPTH = 64)
  int* fillc = died + 64;                                                 // [64]: next record slot per sub-list the wave's slice touches
  const int ntop = sc.num_top;
  const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + wib);  // (the compiler cannot see that threadIdx.x >> 6 is wave-uniform)
  const int lane = lane_id();
  died[lane] = 0;
  const uint64_t t_begin = __builtin_amdgcn_s_memrealtime();  // (100 MHz, whatever the shader clock does)
  // Which queue the wave serves, as which of how many: W / Q waves per queue, or — once a batch has been measured — the
  // queue's share of the W waves by the time its waves took in the previous batch (ptd::Queues::deal).
  int q = wave % qs.Q, r = wave / qs.Q, wq = qs.W / qs.Q;
  if (qs.deal != nullptr && qs.deal[qs.Q] == qs.W) {
    int at_or_before = 0;  // first[] is strictly increasing: the queues whose first wave is <= this wave
    for (int e0 = 0; e0 < qs.Q; e0 += 64) at_or_before += (int)__popcll(ballot(e0 + lane < qs.Q && qs.deal[e0 + lane] <= wave));
    q = at_or_before - 1;
    const int first = qs.deal[q];
    r = wave - first, wq = qs.deal[q + 1] - first;
  }
  const Retire rt = retire_of(ret, q);
  const size_t per_depth = (size_t)qs.Q * qs.cnt_stride;
  const int64_t qbase = (int64_t)q * qs.cap;
  // The queue's depth-1 rays: k_primary's sub-lists (k, rho), e = k * wq0 + rho, concatenated in that order (their survivor
  // counts — the low words of sub[e] — are final before this launch).  A wave takes pieces [p * ps, (p + 1) * ps) of the global
  // ranks and walks each front to back with a cursor: the sub-list the next rank lies in, where that sub-list starts in the queue's
  // path region, and where the records of its paths go — the slots from retirees(e) + i on in sub-region e for the paths from
  // index i on (RetireBuf).
  const int wq0 = rt.wq0, ne = b.K * wq0;
  const int my_nq = queue_share(b, qs, q).my_nq, quo = my_nq / wq0, rem = my_nq % wq0;
  auto count_of = [&](int e) { return e < ne ? (int)(uint32_t)rt.sub[e] : 0; };
  // Every wave of the queue scans sub[] for the total and for where its slice begins.  With a small tile that is thousands of
  // words (an eighth of 1080p: 195 iterations x 20 residues): the words are fetched four 64-entry chunks ahead of their
  // reductions, and the chunk sums stay in a register (lane i: chunk i) so that the second scan starts at the slice's chunk.
  int total = 0, csum = 0;
  const bool chunk_sums = ne <= 64 * 64;
  for (int e0 = 0; e0 < ne; e0 += 256) {
    int n4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) n4[j] = count_of(e0 + 64 * j + lane);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int sum;
      (void)wave_prefix6(n4[j], sum);
      total += sum;
      if (lane == (e0 >> 6) + j) csum = sum;
    }
  }
  if (r == 0 && lane == 0) cnt[per_depth * 1 + (size_t)q * qs.cnt_stride] = total;  // statistics: rays traced at depth 1
  // The queue's ranks [0, total) are cut into pieces of `ps` paths.  Piece r is the wave's own; the pieces from wq on go, in
  // order, to whichever wave of the queue has finished what it had (a counter per queue behind ptd::Queues::deal, zeroed by
  // k_count_stats): equal numbers of depth-1 rays are not equal work — a piece's paths come from a few dozen pixel chunks, and
  // how long a path lives depends on where it starts (a wave of the equal-slices form was resident for 74 % of the launch on
  // average: SQ_WAVE_CYCLES against SQ_BUSY_CYCLES).  A wave streams its pieces through ONE set of lanes: it moves on to the
  // next piece while the last paths of the previous one are still in flight.  Which wave traces a path changes no sample.
  // Piece sizes fall: the pieces of level l = p / wq hold ps0 (1 - 1 / P)^l paths (not fewer than a refill's worth; tests: fewer),
  // ps0 = total / (wq P), P = paths_pieces, so that the levels add up to the queue — large pieces while everybody is busy, small
  // ones at the end, where a launch waits for the last piece (pieces of one size left half a piece of idle time per wave).
  const int pieces_per_wave = qs.deal != nullptr && chunk_sums && (b.paths_pieces & 0xffff) > 1 ? (b.paths_pieces & 0xffff) : 1;
  const int ps_min = b.paths_pieces >> 16;
  const int ps0 = max((total + wq * pieces_per_wave - 1) / (wq * pieces_per_wave), ps_min);
  int ce = ne, cstart = total, ccnt = 0, clist = 0;  // cursor (wave-uniform): sub-list e = k * wq0 + rho; nothing to stream leaves it at the end
  int cord = 0;  // sub-list visits of the cursor so far
  // A window of 64 consecutive sub[] words in registers (lane l holds sub[win0 + l]): the cursor reads counts and retiree numbers
  // with v_readlane instead of a dependent global load per sub-list.  With a whole 1080p frame a wave's piece touches a handful of
  // sub-lists of ~1000 paths; with an eighth of it (eight GPUs, ~190 iterations per batch) dozens of sub-lists of ~25, a fifth of
  // them empty, and two memory round trips per sub-list were a fifth of the kernel's time.
  int win0 = -64;  // nothing loaded yet: no index e >= 0 lies in [-64, 0)
  unsigned long long wsub = 0ull;
  auto window_to = [&](int e) {  // make sub[e] available (wave-uniform e)
    if (e < win0 || e >= win0 + 64) {
      win0 = e;
      wsub = (e + lane < ne) ? rt.sub[e + lane] : 0ull;
    }
  };
  auto sub_word = [&](int e) -> unsigned long long {  // sub[e] for a wave-uniform e inside the window
    const uint32_t lo32 = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)wsub, e - win0);
    const uint32_t hi32 = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(wsub >> 32), e - win0);
    return ((unsigned long long)hi32 << 32) | lo32;
  };
  window_to(0);
  // The records of the paths a wave takes out of ONE visit of a sub-list fill a contiguous range of its sub-region, whatever
  // order they arrive in, so the wave appends them in order of retirement through a counter per visit in LDS (fillc, a ring of
  // 64: lanes of one sub-list that die together get consecutive slots, as the per-iteration counters of the earlier layouts
  // gave).  A lane remembers the visit number of its path; the refill below keeps the ring from lapping a path in flight.
  auto cursor_bases = [&](int first_rank) {
    const int ec = min(ce, ne - 1), ck = ec / wq0, crho = ec - ck * wq0;  // (wave-uniform: scalar instructions)
    clist = ck * rt.seg_cap + sub_offset(quo, rem, crho) * 64;
    window_to(ec);
    const int crec = clist + (int)(uint32_t)(sub_word(ec) >> 32);
    if (lane == 0) fillc[cord & 63] = crec + (first_rank - cstart);  // where this wave's first record of the visit goes
  };
  // put the cursor on the sub-list that rank `at_rank` (< total) lies in: a new visit
  auto seek = [&](int at_rank) {
    int cum = 0, e_begin = 0;
    if (chunk_sums) {
      int all;
      const int cbefore = wave_prefix6(csum, all);
      const unsigned long long here = ballot(csum > 0 && cbefore <= at_rank && at_rank < cbefore + csum);
      if (here) {
        const int l = __builtin_ctzll(here);
        e_begin = l * 64, cum = __builtin_amdgcn_readlane(cbefore, l);
      }
    }
    for (int e0 = e_begin; e0 < ne; e0 += 64) {
      const int n = count_of(e0 + lane);
      int sum;
      const int before = wave_prefix6(n, sum);
      const unsigned long long here = ballot(n > 0 && cum + before <= at_rank && at_rank < cum + before + n);
      if (here) {
        const int l = __builtin_ctzll(here);
        ce = e0 + l, cstart = cum + __builtin_amdgcn_readlane(before, l), ccnt = __builtin_amdgcn_readlane(n, l);
        break;
      }
      cum += sum;
    }
    ++cord;
    cursor_bases(at_rank);
  };
  // Path index (inside the queue's region) of the rays of global rank `rank`, for the lanes that `want` one (consecutive ranks in
  // lane order), and the visit their retirement records are counted under.  Ranks only grow inside a piece, so the cursor only
  // moves forward; it stops after 31 new visits (sub-lists of a path or two: tiles of a few pixels per wave) — the lanes behind
  // that are served by a later refill.  Returns whether the lane was served.  Wave-uniform control flow.
  auto assign = [&](bool want, int rank, int& at, int& rs) -> bool {
    bool pending = want;
    const int cord0 = cord;
    while (true) {
      const bool in = pending && rank < cstart + ccnt;
      if (in) at = clist + (rank - cstart), rs = cord;
      pending = pending && !in;
      if (!ballot(pending) || ce >= ne || cord - cord0 >= 31) break;
      cstart += ccnt;  // on to the next sub-list that holds anything: the first non-empty one behind ce in the window, else the window moves on
      ccnt = 0;
      int nxt = ce + 1;
      while (nxt < ne) {
        window_to(nxt);
        const unsigned long long holds = ballot((uint32_t)wsub != 0u) & (~0ull << (nxt - win0));
        if (holds) {
          nxt = win0 + (int)__builtin_ctzll(holds);
          ccnt = (int)(uint32_t)sub_word(nxt);
          break;
        }
        nxt = win0 + 64;
      }
      ce = min(nxt, ne);
      ++cord;
      cursor_bases(cstart);
    }
    return want && !pending;
  };
  const uint32_t s_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_offset(slots));
  const uint32_t s16 = s_base + (uint32_t)lane * 16u, s4 = s_base + 2048u + (uint32_t)lane * 4u;
  // issue the transfer of path record `i` of the queue into this lane's slot
  const ptd::Word4 *in0 = uniform_ptr(in.r + qbase), *in1 = uniform_ptr(in.r + in.stride + qbase);
  const float* in2 = uniform_ptr(reinterpret_cast<const float*>(in.r + 2 * in.stride) + 2 * qbase);
  PathRec nx;  // !SLOTS: the lane's next record, in registers
  nx.o = nx.d = nx.c = mk(0.f, 0.f, 0.f), nx.tag = PathTag{0, 0u, 0};
  int nx_rs = 0;  // !SLOTS: ... and its visit
  int* slot_rs = reinterpret_cast<int*>(slots + 2560) + lane;  // SLOTS: the visit of the record waiting in the lane's slot
  auto fetch = [&](int at, int rs) {
    if constexpr (SLOTS) fetch_record_to_lds(in0, in1, in2, at, s_base), *slot_rs = rs;
    else nx = path_load(in, qbase + at), nx_rs = rs;
  };
  // lane state
  f3 o = mk(0.f, 0.f, 0.f), d = o, c = o;
  int slot = 0, depth = 1, mark = 0, rslot = 0;  // rslot: the visit the path was taken in; from its death on: its record slot
  uint32_t phash = 0u;
  bool valid = false, fresh = false, owes = false;  // owes: the lane's path died and its retirement record is not stored yet
  bool has_next = false;                            // a record is waiting in the lane's slot (on its way there)
  int streamed = 0, p_hi = 0;  // the current piece: records [.., streamed) handed out, the piece ends at p_hi; -1: no piece left for this wave
  while (true) {
    // ── on to the wave's next piece, once the current one is handed out: its own, then whatever the queue's counter gives ──
    if (streamed >= p_hi && p_hi >= 0) {
      int nextp = -1;
      if (cord == 0) nextp = r;  // (every piece starts with a visit)
      else if (ps0 * wq < total) {  // (level 0 does not cover the queue: there is a counter)
        int v = 0;
        if (lane == 0) v = atomicAdd(&qs.deal[2 * qs.Q + 2 + q], 1);
        nextp = wq + __builtin_amdgcn_readfirstlane(v);
      }
      int start = total, sz = ps0;
      if (nextp >= 0) {
        start = 0;
        int level = nextp / wq;
        const int idx = nextp - level * wq;
        for (; level > 0 && start < total; --level) start += wq * sz, sz = max(sz - sz / pieces_per_wave, ps_min);
        start += idx * sz;
      }
      if (start < total) {
        streamed = start, p_hi = min(start + sz, total);
        seek(streamed);
      } else {
        p_hi = -1;
      }
    }
    // ── refill: dead lanes take the record waiting in their slot; the slot gets the next record of the piece ──
    const bool take = !valid && has_next;
    if (ballot(take || owes || (!valid && !has_next && streamed < p_hi))) {
      v4f w0, w1;
      float cz;
      int nslot, nrs;
      if constexpr (SLOTS) {
        // everything outstanding here (last refill's transfers, last refill's retirement stores) is a whole iteration old
        asm volatile(
            "s_waitcnt vmcnt(0)\n\t"
            "ds_read_b128 %0, %5\n\t"
            "ds_read_b128 %1, %5 offset:1024\n\t"
            "ds_read_b32 %2, %6\n\t"
            "ds_read_b32 %3, %6 offset:256\n\t"
            "ds_read_b32 %4, %6 offset:512\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(w0), "=&v"(w1), "=&v"(cz), "=&v"(nslot), "=&v"(nrs)
            : "v"(s16), "v"(s4)
            : "memory");
      } else {
        w0 = v4f{nx.o.x, nx.o.y, nx.o.z, nx.d.x}, w1 = v4f{nx.d.y, nx.d.z, nx.c.x, nx.c.y}, cz = nx.c.z, nslot = nx.tag.slot, nrs = nx_rs;
      }
      if (owes) {  // the record of the path that died in this lane (its colour and sample id are still here)
#ifndef PT_ABL_NO_RETIRE
        rt.rec[rslot] = ptd::Word4{c.x, c.y, c.z, __int_as_float(slot & ((1 << b.slot_shift) - 1))};
#endif
        owes = false;
      }
      if (take) {
        o = mk(w0.x, w0.y, w0.z), d = mk(w0.w, w1.x, w1.y), c = mk(w1.z, w1.w, cz);
        slot = nslot, rslot = nrs;
        phash = utilhash((uint32_t)global_pixel(b, nslot & ((1 << b.slot_shift) - 1)));
        depth = 1;
        valid = fresh = true;
        has_next = false;
      }
      // every lane without a waiting record gets the next one of the piece — unless a path in flight was taken 32 or more visits
      // ago: its counter in the ring of 64 must not be handed to another visit (the cursor moves by at most 31 per refill)
#ifdef PT_EXP_NO_LAP_GUARD  // timing experiment only (records of tiny sub-lists may collide)
      const bool lapping = false;
#else
      const bool lapping = ballot(valid && cord - rslot >= 32) != 0ull;
#endif
      const bool empty = !has_next && !lapping;  // takers (just emptied) and lanes that found nothing at an earlier refill
      const unsigned long long em = ballot(empty);
      const int rank = rank_in(em);
      const bool more = empty && streamed + rank < p_hi;
      int at = 0, rs = 0;
      const bool served = assign(more, streamed + rank, at, rs);
      if (served) {
        has_next = true;
        fetch(at, rs);
      }
      streamed += (int)__popcll(ballot(served));
    }
    // (no valid lane but records on their way, at the start of the wave: the rest of the round finds nothing to do — a `continue`
    // here costs six VGPRs)
    if (!ballot(valid || has_next) && p_hi < 0) break;  // every path of the wave's pieces has retired
    bool ready;
    if constexpr (MODE == 0) {
      // ── search: box tests + appends for the lanes with a new ray; full chunks as the ring fills ──
      if (ballot(fresh)) {
        if constexpr (PT_TOP_SCALAR != 0) paths_search(cy, (cfloat4*)(uintptr_t)sc.top_b, tword, ntop, geoms, o, d, fresh, lane, mark);
        else paths_search(cy, top, tword, ntop, geoms, o, d, fresh, lane, mark);
      }
      // ── which lanes are resolved?  Too few, with candidates pending: run them as a partial chunk ──
      ready = valid && (cy.processed - mark) >= 0;
      if (cy.count > 0 && __popcll(ballot(ready)) < kPathsMinReady) {
        paths_chunk(cy, cy.count, lane, o, d, tword, geoms);  // count < 64 here
        ready = valid;
      }
    } else {
      // ── search of every live lane (all of them have a new ray), then all its candidates ──
      cb.best[lane] = kNoHit;
      if constexpr (MODE == 2) grid_search<1>(cb, cr, sc, nodes, geoms, o, d, valid, lane, 0);
      else {
        PT_STAT(14, 1);
        PT_STAT(15, __popcll(ballot(valid)));
        carry_search<true, 1>(cb, top, ntop, nodes, geoms, o, d, valid, lane, 0, sc.cull_margin, sc.top_xor);
      }
      while (cb.count > 0) carry_chunk<false, 1, false, MODE == 2>(cb, min(64, cb.count), lane, nodes, geoms);
      ready = valid;
    }
    fresh = false;
    // ── shade the resolved lanes ──
    {
      const unsigned long long best = cy.best[lane];
      const bool hit = (uint32_t)(best >> 32) != 0x7f7fffffu;
      ShadeIO s;
      s.o = mk(0.f, 0.f, 0.f);
      s.d = d;
      s.c = c;
      s.alive = false;
      Bounce bo;
      bo.kind = 0;
      f3 hn = mk(0.f, 0.f, 0.f), hp = mk(0.f, 0.f, 0.f);
      const int k = (int)((uint32_t)slot >> b.slot_shift);
      if (ready) {
        float ht = -1.0f;
        int hmat = 0;
        if (hit) {
          ht = __uint_as_float((uint32_t)(best >> 32));
          const float* rr = cy.rec + lane;
          hn = mk(rr[0 * 64], rr[1 * 64], rr[2 * 64]);
          hp = mk(rr[3 * 64], rr[4 * 64], rr[5 * 64]);
          if constexpr (MODE == 0) {
            hmat = lmat[(uint32_t)best & 63u];
          } else {
            const ptd::Geom* G = geoms + nodes[(uint32_t)best].geom;
            hmat = G->material;
            if (MODE == 2) hn = finish_normal(G, hn);  // the grid's chunks leave the normal to the winner (carry_chunk, LEAN)
          }
        }
        const uint32_t ih = he > 0 ? ihash[(depth - 1) * he + k] : iter_hash(b.iter_first + k, depth);
        bo = shade_decide(mats, b.trace_depth, depth, ih ^ phash, ht, hmat, s);
      }
      const bool alive = ready && s.alive, dead = ready && !s.alive;
      if (dead) atomicAdd(&died[depth & 63], 1);  // statistics: rays traced at depth d = paths retired at depth >= d
      if (dead) rslot = atomicAdd(&fillc[rslot & 63], 1);  // lanes of one visit get consecutive record slots
      if (alive) shade_bounce(bo, hn, hp, s);
      if (ready) {
        c = s.c;
        if (alive) o = s.o, d = s.d, depth += 1, fresh = true;
        else valid = false, owes = true;
      }
    }
  }
  // statistics: rays traced at depth d >= 2 = this wave's paths retired at depth >= d (row 1 holds the queue's input count already)
  if (lane == 0) {
    int reached = 0, rays = 0;
    for (int dd = min(b.trace_depth - 1, 63); dd >= 2; --dd) {
      reached += died[dd];
      rays += reached;
      if (reached) atomicAdd(&cnt[per_depth * dd + (size_t)q * qs.cnt_stride], reached);
    }
    // What this queue's paths cost, for the next batch's deal: the TIME its waves spent on them (the waves of a queue finish together
    // — they share its pieces — so the sum is waves x the queue's finishing time, and dealing in proportion to it moves the
    // finishing times together whatever a ray costs where).  Rays traced were the first measure: an eighth of 1080p kept waves
    // resident for 73 % of the launch with it.
    rays += reached + died[1];  // (the wave's rays: every path it took has retired somewhere)
    const int ticks = (int)min((uint64_t)(__builtin_amdgcn_s_memrealtime() - t_begin) >> 6, (uint64_t)0x3ffff);  // units of 0.64 us; a queue's sum stays below 2^31
    if (qs.deal != nullptr && rays) atomicAdd(&qs.deal[qs.Q + 1 + q], ticks), atomicAdd(&qs.deal[3 * qs.Q + 2 + q], rays);
  }
}

// test-only stage: explicit (iter, pixel) per path, in-place, no compaction
__global__ __launch_bounds__(kBlock) void k_shade_stage(SceneTables sc, int trace_depth, int depth, int n,
                                                        const int32_t* __restrict__ iter,
                                                        const int32_t* __restrict__ pixel, ptd::HitBuf hits,
                                                        ptd::PathBuf paths, int32_t* __restrict__ alive) {
  extern __shared__ float4 lds_raw[];
  stage16(lds_raw, sc.mats, sc.num_mats * (int)sizeof(ptd::Mat));
  __syncthreads();
  const ptd::Mat* mats = reinterpret_cast<const ptd::Mat*>(lds_raw);
  const int64_t HS = hits.stride;
  for (int at = blockIdx.x * blockDim.x + threadIdx.x; at < n; at += gridDim.x * blockDim.x) {
    const PathRec pr = path_load(paths, at);
    ShadeIO s;
    s.o = pr.o;
    s.d = pr.d;
    s.c = pr.c;
    s.alive = false;
    const Bounce bo = shade_decide(mats, trace_depth, depth, iter_hash(iter[at], depth) ^ utilhash((uint32_t)pixel[at]), hits.t[at], hits.mat[at], s);
    // the stage reports the bounce ray whenever one is sampled (also at the last depth), like the reference
    if (bo.kind) shade_bounce(bo, mk(hits.n[at], hits.n[HS + at], hits.n[2 * HS + at]),
                              mk(hits.p[at], hits.p[HS + at], hits.p[2 * HS + at]), s);
    path_store(paths, at, s.o, s.d, s.c, pr.tag);
    alive[at] = s.alive ? 1 : 0;
  }
}

#include "pt_output.inc"
#include "pt_launch.inc"
const KernelApi kApi = {
#if PT_ARITH == 0
    "exact",
#elif PT_ARITH == 1
    "fma",
#else
    "fast",
#endif
    launch_generate, launch_primary, launch_intersect, launch_shade, launch_collect, launch_count_stats,
    launch_preview, launch_save_u8, launch_shade_stage, lds_table_limit, resident_blocks_per_cu, launch_ieee_check, launch_paths,
    md::kFastSlab ? 1 : 0};

}  // namespace
}  // namespace PT_NS

const KernelApi* PT_API_FN() { return &PT_NS::kApi; }

#if defined(PT_WALK_STATS) && PT_ARITH == 2
}  // namespace ptk
// diagnostic builds only: the fast build's counters since the last call (read and reset)
extern "C" int pt_debug_walk_stats(unsigned long long* out) {
  unsigned long long zero[16] = {0};
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(ptk::arith_fast::g_walk_stats), sizeof(zero)) != hipSuccess) return -1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(ptk::arith_fast::g_walk_stats), zero, sizeof(zero)) != hipSuccess) return -1;
  return 16;
}
namespace ptk {
#endif

}  // namespace ptk
