// pt_api.cpp — C ABI (include/pt_amd.h) and device-state owner of the MI355X
// wavefront path tracer.  Host C++ calling the HIP runtime directly; replaces the
// reference's pathtraceInit / pathtrace / pathtraceFree (src/pathtrace.cu:446-653).
//
// Differences from the reference's host loop by design:
//   * no host<->device synchronisation inside an iteration (the reference does ~38),
//     no per-iteration malloc/free, no per-iteration D2H frame copies or printf;
//   * K iterations are traced as one wavefront batch so each launch has enough rays;
//   * kernel sizes are fixed (persistent grid), live counts stay on the device.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/pt_amd.h"
#include "pt_device.h"
#include "pt_internal.h"
#include "pt_kernels.h"
#include "pt_scene.h"

namespace {

#ifdef PT_ABLATE
constexpr bool kAblateBuild = true;   // tools/pmc_ablate.sh A/B library: PtOptions.debug_flags bits 0-3 honoured
#else
constexpr bool kAblateBuild = false;  // release library: pt_init rejects them
#endif

std::string g_err;
constexpr int kGridNodes = 600;        // scenes from this many BVH nodes on are candidates for the uniform grid (build_grid, choose_traversal); the ladder scene of 500 primitives (999 nodes): scan 3.9 k, grid 4.7 k Msamples/s, 156 primitives (311 nodes): 5.5 / 5.3
constexpr int kTightNodes = 64;        // scenes from this many BVH nodes on test sphere leaves against the ellipsoid's box (sphere_tight_box); the
                                       // reference's own scenes (cornell.txt: 13 nodes) keep the reference's boxes
}  // namespace
int pt_fail(const char* fmt, ...) {  // pt_internal.h: sets pt_last_error(), returns -1
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return -1;
}
namespace {
#define fail pt_fail
#define HIP_OK(expr)                                                                                   \
  do {                                                                                                 \
    hipError_t e_ = (expr);                                                                            \
    if (e_ != hipSuccess) return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

struct EventPair {
  hipEvent_t a, b;
};

}  // namespace

// One renderer instance = one device, one stream, one tile of the framebuffer.  The reference keeps this state in
// file-scope statics (pathtrace.cu:446-456); here it is an object so that one process can drive several GPUs
// (pt_group_*, pt_group.cpp) — the old single-instance entry points act on a default context.
struct PtContext {
  int device = 0;
  hipStream_t stream = nullptr;
  const ptk::KernelApi* k = nullptr;  // kernels of the selected arithmetic mode
  int arith = 0;
  // scene
  std::vector<PtGeom> geoms;
  std::vector<PtMaterial> mats;
  PtCamera cam{};
  ptd::Camera dcam{};
  int depth = 0;
  // tile / batch geometry
  int N = 0, pixel_begin = 0, K = 1;
  int slot_shift = 0;  // BatchInfo::slot_shift
  int stripe = 0, stripe_stride = 0;
  int num_cus = 0, grid = 0;  // grid: widest persistent grid (stats / test stages)
  int grid_gen = 0, grid_isect = 0, grid_shade = 0;
  ptd::Queues qs{};
  int64_t stride = 0;  // plane stride = Q*cap
  // device memory
  std::vector<void*> allocs;
  int64_t device_bytes = 0;
  ptd::Node* d_nodes = nullptr;
  ptd::Geom* d_geoms = nullptr;
  ptd::Mat* d_mats = nullptr;
  ptd::TopEntry* d_top = nullptr;
  // the bounce kernels' box tables (SceneTables::*_b): the same buffers, except for the fast build (centre / half extent copies)
  ptd::Node* d_nodes_b = nullptr;
  ptd::TopEntry* d_top_b = nullptr;
  ptd::Node* d_grid_items_b = nullptr;
  int num_nodes = 0, num_top = 0;
  float root_min[3] = {0, 0, 0}, root_max[3] = {0, 0, 0};
  float cull_margin = 0.f;
  unsigned long long top_xor = 0;  // SceneTables::top_xor
  int lds_table_bytes = -1;        // SceneTables::lds_table_bytes
  // uniform grid over the leaf boxes (build_grid; SceneTables::grid_*), large scenes where it beats the BVH scan (choose_traversal)
  uint32_t* d_grid_start = nullptr;
  size_t grid_guard = 0;  // empty cells in front of (and behind) the cell table proper
  ptd::Node* d_grid_items = nullptr;
  int grid_res[3] = {0, 0, 0};
  float grid_min[3] = {0, 0, 0}, grid_cs[3] = {0, 0, 0}, grid_inv_cs[3] = {0, 0, 0}, grid_pad = 0.f;
  bool have_grid = false;
  // Grids of other resolutions built next to the host's first choice; choose_traversal() times them all, keeps the fastest
  // in the fields above and frees the rest.
  struct GridAlt {
    uint32_t* d_start;
    size_t guard;
    ptd::Node* d_items;
    ptd::Node* d_items_b;  // == d_items unless the build wants centre / half extent
    int res[3];
    float gmin[3], cs[3], inv_cs[3], pad;
    size_t bytes;
  };
  std::vector<GridAlt> grid_alts;
  int tight_leaves = 0;  // sphere leaves with a tightened traversal box (tighten_sphere_leaves)
  bool grid_enabled = true;      // the outcome of choose_traversal()
  float probe_ms[2] = {0.f, 0.f};  // a few iterations with the BVH scan / with the (fastest) grid, as timed by choose_traversal()
  int cap_bpc = 8;
  bool legacy = false;
  int debug_flags = 0;
  bool fuse_primary = true, fuse_bounces = true;
  bool aa_jitter = false;
  bool has_triangles = false;  // SceneTables::has_triangles
  int grid_primary = 0, grid_paths = 0;
  ptd::PathBuf buf[2]{};
  ptd::HitBuf hits{};
  ptd::RetireBuf ret{};  // retirement records + fill levels (pt_device.h)
  float* d_image = nullptr;
  uint8_t* d_rgb8 = nullptr;  // lazily allocated output of pt_ctx_save_u8
  int32_t* d_cnt = nullptr;
  unsigned long long* d_stats = nullptr;
  // timing
  bool time_kernels = false;
  std::vector<EventPair> free_events, pending_isect, pending_render;
  double isect_ms = 0, render_ms = 0;
  int64_t isect_launches = 0;
  int64_t samples = 0;
  // A batch that failed half-way (a launch or an event call returned an error) leaves counters and record regions in an
  // undefined state: the context refuses further renders instead of appending past them.
  bool failed = false;
};

namespace {
using Ctx = PtContext;
Ctx* g_default = nullptr;  // the instance behind pt_init / pt_render / pt_free

template <typename T>
int dalloc(Ctx& g, T** out, size_t count) {
  void* p = nullptr;
  size_t bytes = std::max<size_t>(count * sizeof(T), 16);
  hipError_t e = hipMalloc(&p, bytes);
  if (e != hipSuccess) return fail("hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
  g.allocs.push_back(p);
  g.device_bytes += (int64_t)bytes;
  *out = reinterpret_cast<T*>(p);
  return 0;
}

int get_events(Ctx& g, EventPair* ev) {
  if (!g.free_events.empty()) {
    *ev = g.free_events.back();
    g.free_events.pop_back();
    return 0;
  }
  HIP_OK(hipEventCreate(&ev->a));
  HIP_OK(hipEventCreate(&ev->b));
  return 0;
}
int resolve_events(Ctx& g) {  // requires the stream to be idle
  for (auto& e : g.pending_isect) {
    float ms = 0;
    HIP_OK(hipEventElapsedTime(&ms, e.a, e.b));
    g.isect_ms += ms;
    g.isect_launches++;
    g.free_events.push_back(e);
  }
  g.pending_isect.clear();
  for (auto& e : g.pending_render) {
    float ms = 0;
    HIP_OK(hipEventElapsedTime(&ms, e.a, e.b));
    g.render_ms += ms;
    g.free_events.push_back(e);
  }
  g.pending_render.clear();
  return 0;
}

// Re-emit the reference-order BVH (node, left, right links) in visiting order with
// skip links (see ptd::Node).  Visiting order of the reference's stack walk is:
// node, then its RIGHT subtree, then its LEFT subtree (pathtrace.cu:321-322).
void thread_bvh(const std::vector<PtBVHNode>& in, int idx, std::vector<ptd::Node>& out, std::vector<int>& where) {
  const PtBVHNode& n = in[idx];
  const size_t self = out.size();
  where[idx] = (int)self;
  ptd::Node t{};
  std::memcpy(t.bmin, n.bmin, 12);
  std::memcpy(t.bmax, n.bmax, 12);
  t.geom = n.left < 0 ? n.geomIndex : -1;
  out.push_back(t);
  if (n.left >= 0) {
    thread_bvh(in, n.right, out, where);
    thread_bvh(in, n.left, out, where);
  }
  out[self].skip = (int32_t)out.size();
}

// Flatten the top of the tree into at most ptk::kMaxTop entries (see ptd::TopEntry): start from
// the root and keep splitting the inner entry that covers the most nodes.  The entries are emitted in
// path-code order (0 = left child, 1 = right child, root choice in the most significant bit); when
// the cut is a complete level (every entry at the same depth, a power of two of them) `top_xor`
// receives, per direction-sign octant, the XOR mask that turns list order into near-first order
// (pt_kernels.hip permute_xor), otherwise 0.
void build_top(const std::vector<PtBVHNode>& ref, const std::vector<ptd::Node>& thr, const std::vector<int>& where,
               const std::vector<PtGeom>& geoms, std::vector<ptd::TopEntry>& top, unsigned long long* top_xor) {
  struct Cut {
    int ref_idx;
    uint32_t code;
    int len;
  };
  std::vector<Cut> cut{{0, 0u, 0}};
  auto span = [&](int ref_idx) { return thr[where[ref_idx]].skip - where[ref_idx]; };
  const char* te = getenv("PT_TOP_ENTRIES");  // experiment knob: a smaller cut (scenes with subtrees only: the LDS-table kernels need every leaf in the list)
  const int want = te ? std::max(1, std::min(atoi(te), ptk::kMaxTop)) : ptk::kMaxTop;
  const int max_top = ((int)(ref.size() + 1) / 2 <= ptk::kMaxTop && !getenv("PT_LDS_TABLE_KB")) ? ptk::kMaxTop : want;
  while ((int)cut.size() < max_top) {
    int best = -1;
    for (size_t i = 0; i < cut.size(); ++i)
      if (ref[cut[i].ref_idx].left >= 0 && cut[i].len < 24 && (best < 0 || span(cut[i].ref_idx) > span(cut[best].ref_idx)))
        best = (int)i;
    if (best < 0) break;
    const Cut c = cut[best];
    const PtBVHNode n = ref[c.ref_idx];
    cut[best] = Cut{n.left, c.code << 1, c.len + 1};
    cut.push_back(Cut{n.right, (c.code << 1) | 1u, c.len + 1});
  }
  std::sort(cut.begin(), cut.end(), [](const Cut& a, const Cut& b) {
    return ((uint64_t)a.code << (32 - a.len)) < ((uint64_t)b.code << (32 - b.len));
  });
  top.clear();
  for (const Cut& c : cut) {
    const PtBVHNode& n = ref[c.ref_idx];
    ptd::TopEntry e{};
    std::memcpy(e.bmin, n.bmin, 12);
    std::memcpy(e.bmax, n.bmax, 12);
    e.idx = where[c.ref_idx];
    e.link = n.left < 0 ? -1 - geoms[n.geomIndex].type : thr[where[c.ref_idx]].skip;
    top.push_back(e);
  }
  *top_xor = 0;
  const int levels = cut[0].len;
  bool complete = levels >= 1 && levels <= 5 && (int)cut.size() == (1 << levels);
  for (const Cut& c : cut) complete = complete && c.len == levels;
  if (!complete) return;
  // axis and polarity of each level, taken from the node on the all-left path
  int axis[5] = {0, 0, 0, 0, 0};
  bool left_lower[5] = {true, true, true, true, true};
  int at = 0;
  for (int l = 0; l < levels; ++l) {
    const PtBVHNode &L = ref[ref[at].left], &R = ref[ref[at].right];
    float sep = -1.f;
    for (int a = 0; a < 3; ++a) {
      const float d = fabsf((L.bmin[a] + L.bmax[a]) - (R.bmin[a] + R.bmax[a]));
      if (d > sep) sep = d, axis[l] = a;
    }
    left_lower[l] = (L.bmin[axis[l]] + L.bmax[axis[l]]) <= (R.bmin[axis[l]] + R.bmax[axis[l]]);
    at = ref[at].left;
  }
  for (int oct = 0; oct < 8; ++oct) {
    uint32_t m = 0;
    for (int l = 0; l < levels; ++l) {
      const bool negative = (oct >> axis[l]) & 1;
      const bool near_is_right = left_lower[l] == negative;  // positive direction: the lower child is nearer
      if (near_is_right) m |= 1u << (levels - 1 - l);
    }
    *top_xor |= (unsigned long long)m << (8 * oct);
  }
}

// normalize(vec3(invTranspose * vec4(n, 0))) in GLM operation order (same expressions as mulMV /
// normalize in pt_kernels.hip; this file is compiled with -ffp-contract=off, sqrtf and / are IEEE on
// the host as on the device), for the 7 object-space normals a cube test can produce.
void box_normal_table(const float invT12[12], float out[7][4]) {
  for (int code = 0; code < 7; ++code) {
    float n[3] = {0.f, 0.f, 0.f};
    if (code > 0) n[(code - 1) / 2] = ((code - 1) & 1) ? 1.0f : -1.0f;
    const float* m = invT12;
    float v[3];
    for (int r = 0; r < 3; ++r) v[r] = (m[0 + r] * n[0] + m[3 + r] * n[1]) + (m[6 + r] * n[2] + m[9 + r] * 0.0f);
    const float inv = 1.0f / sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
    out[code][0] = v[0] * inv, out[code][1] = v[1] * inv, out[code][2] = v[2] * inv, out[code][3] = 0.f;
  }
}

void pack_rows(const float m16[16], float out12[12]) {
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 3; ++r) out12[c * 3 + r] = m16[c * 4 + r];
}

// Tighter leaf boxes for spheres (large scenes only; result-neutral).  The reference's leaf box is the box of the
// transformed unit CUBE (pathtrace.cu:36-50), which for a rotated sphere / ellipsoid is up to sqrt(3) wider per axis than
// the primitive: rays pass the box, reach sphereIntersectionTest (intersections.h:102-144) and miss.  Our structures may
// skip such a leaf exactly when its primitive test cannot report a hit, so the box our traversal tests is
//     reference leaf box  INTERSECTED WITH  box of the ellipsoid { x : |A (x - c)| <= rho },  A = the geom's inverse transform,
// where rho > 0.5 covers everything the reference's float arithmetic can still call a hit.  That test takes
// radicand = v*v - (|ro|^2 - 0.25) >= 0 with ro = A o + t, rd = normalize(A d), v = ro . rd in float (u = 2^-24):
//   * the two dot products, the square and the two subtractions:  <= 18 u |ro|^2,         |ro| <= |A|_F * D  =: D_obj
//   * rounding of ro itself (4 products per component, cancelling against t when the object is far from the origin):
//     |d ro| <= 14 u (|A|_F * M + |t|) =: 14 u M_obj, which moves the squared distance to the axis by <= 2 * 0.5 * |d ro|
//   * rounding of rd (direction error <= 5 u cond(A)), which moves it by <= 5 u cond(A) * D_obj
// with D = the largest distance from a ray origin to the sphere's centre and M = the largest coordinate magnitude of a ray
// origin; ray origins of the renderer lie in the scene bounds (hit points) or at the camera.  rho^2 = 0.25 + 4 x that sum
// (safety factor 3.5 on the dominant term), the box is rounded outward and padded by 32 float-eps of M for the slab test's
// own rounding.  The tightened box lies inside the reference's, so a ray that passes it passes the reference's box and
// every ancestor (the slab test is monotone under inclusion, DESIGN.md §9, LABNOTES.md §9.1): the leaves we test are a subset of the
// reference's, and the ones we drop return "no hit" there.  Exact mode stays bit-identical (tests: stress scenes and the
// random-scene fuzz against the oracle, which keeps the reference's boxes); debug_flags 2048 keeps the reference's boxes.
// pt_stage_intersect on such a scene inherits the assumption about ray origins.
// One sphere: ref_lo / ref_hi = the reference's leaf box; returns false when nothing could be tightened.
bool sphere_tight_box(const PtGeom& gm, const double origin_lo[3], const double origin_hi[3], const float ref_lo[3], const float ref_hi[3],
                      float lo[3], float hi[3]) {
  const double u = 1.0 / 16777216.0;
  double M = 0.0;
  for (int a = 0; a < 3; ++a) M = std::max({M, std::fabs(origin_lo[a]), std::fabs(origin_hi[a])});
  M *= std::sqrt(3.0);
  const float* inv = gm.inverseTransform;  // column-major 4x4, m[c*4+r]
  double A[3][3], t[3], B[3][3];
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) A[r][c] = inv[c * 4 + r];
    t[r] = inv[3 * 4 + r];
  }
  const double det = A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) +
                     A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
  if (!(std::fabs(det) > 0.0) || !std::isfinite(det)) return false;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      const int r1 = (c + 1) % 3, r2 = (c + 2) % 3, c1 = (r + 1) % 3, c2 = (r + 2) % 3;  // B = A^-1: cofactor of A[c][r] over det
      B[r][c] = (A[r1][c1] * A[r2][c2] - A[r1][c2] * A[r2][c1]) / det;
    }
  double nA = 0.0, nB = 0.0, nt = 0.0, ctr[3], D = 0.0;
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) nA += A[r][c] * A[r][c], nB += B[r][c] * B[r][c];
    nt += t[r] * t[r];
    ctr[r] = -(B[r][0] * t[0] + B[r][1] * t[1] + B[r][2] * t[2]);
  }
  nA = std::sqrt(nA), nB = std::sqrt(nB), nt = std::sqrt(nt);
  for (int a = 0; a < 3; ++a) {
    const double far = std::max(std::fabs(ctr[a] - origin_lo[a]), std::fabs(ctr[a] - origin_hi[a]));
    D += far * far;
  }
  const double D_obj = nA * std::sqrt(D), M_obj = nA * M + nt;
  const double rho = std::sqrt(0.25 + 4.0 * u * (18.0 * D_obj * D_obj + 14.0 * M_obj + 5.0 * nA * nB * D_obj));
  if (!std::isfinite(rho)) return false;
  const double slab = 32.0 * (double)FLT_EPSILON * M;
  bool changed = false;
  for (int a = 0; a < 3; ++a) {
    const double half = rho * std::sqrt(B[a][0] * B[a][0] + B[a][1] * B[a][1] + B[a][2] * B[a][2]) + slab;
    lo[a] = std::nextafterf((float)(ctr[a] - half), -INFINITY);
    hi[a] = std::nextafterf((float)(ctr[a] + half), INFINITY);
    if (!std::isfinite(lo[a]) || !std::isfinite(hi[a])) lo[a] = ref_lo[a], hi[a] = ref_hi[a];
    lo[a] = std::max(lo[a], ref_lo[a]), hi[a] = std::min(hi[a], ref_hi[a]);
    changed = changed || lo[a] != ref_lo[a] || hi[a] != ref_hi[a];
  }
  return changed && lo[0] < hi[0] && lo[1] < hi[1] && lo[2] < hi[2];
}
int tighten_sphere_leaves(std::vector<ptd::Node>& nodes, const std::vector<PtGeom>& geoms, const double origin_lo[3], const double origin_hi[3]) {
  int tightened = 0;
  for (ptd::Node& n : nodes) {
    if (n.geom < 0 || geoms[n.geom].type != PT_GEOM_SPHERE) continue;
    float lo[3], hi[3];
    if (!sphere_tight_box(geoms[n.geom], origin_lo, origin_hi, n.bmin, n.bmax, lo, hi)) continue;
    std::memcpy(n.bmin, lo, 12);
    std::memcpy(n.bmax, hi, 12);
    ++tightened;
  }
  return tightened;
}
// The region ray origins come from: hit points (inside the scene bounds; the shading offsets are <= 1e-3) and the camera.
void origin_region(const float root_min[3], const float root_max[3], const float cam[3], double olo[3], double ohi[3]) {
  for (int a = 0; a < 3; ++a) {
    const double ext = (double)root_max[a] - (double)root_min[a];
    olo[a] = std::min((double)root_min[a] - 0.01 * ext - 0.01, (double)cam[a]);
    ohi[a] = std::max((double)root_max[a] + 0.01 * ext + 0.01, (double)cam[a]);
  }
}

// Uniform grid over the leaf boxes — the traversal structure of our own for large scenes (SceneTables::grid_*).  The
// reference's median-split BVH (pathtrace.cu:52-111) costs a bounce ray of a 10 000-primitive scene ~60 dependent node
// fetches below the top list; a grid with about one cell per primitive is walked in ~6 cells.  Results cannot change: a
// leaf is tested exactly when the ray passes the leaf's own box (pt_device.h), the grid only has to deliver a superset
// of those leaves, and it does because every leaf is listed in all cells its box grown by `pad` touches — `pad` is orders
// of magnitude above the rounding of the float cell walk.  Built for large scenes whose lists stay moderate; whether
// the renderer walks it or the BVH is measured at init (choose_traversal).
struct GridBuild {
  int res[3];
  float gmin[3], cs[3], inv_cs[3], pad;
  std::vector<uint32_t> start;
  std::vector<ptd::Node> items;
};
// `coord_mag`: the largest coordinate magnitude a ray of this scene is built from (scene bounds and, where known, the
// camera position).  The float cell walk's error grows with eps * |coordinate| and eps * t_in, not with the extent, so
// for a small scene far from the origin (|coordinate| / extent >~ 1e3) 1e-4 * extent alone would sink below the
// walk's rounding; the pad therefore never goes below 64 ulps of that magnitude.
bool build_grid(const std::vector<ptd::Node>& nodes, const std::vector<PtGeom>& geoms, const float root_min[3], const float root_max[3], double coord_mag,
                double density, bool forced, GridBuild& gb) {
  std::vector<int> leaves;
  for (size_t i = 0; i < nodes.size(); ++i)
    if (nodes[i].geom >= 0) leaves.push_back((int)i);
  if (leaves.empty()) return false;
  double ext[3], maxext = 0.0;
  for (int a = 0; a < 3; ++a) maxext = std::max(maxext, (double)root_max[a] - (double)root_min[a]);
  if (!(maxext > 0.0) || !std::isfinite(maxext)) return false;
  for (int a = 0; a < 3; ++a) coord_mag = std::max({coord_mag, std::fabs((double)root_min[a]), std::fabs((double)root_max[a])});
  const double pad = std::max(1e-4 * maxext, 64.0 * (double)FLT_EPSILON * coord_mag);
  if (!(pad < 0.05 * maxext)) return false;  // the scene is a speck at its distance from the origin: cells would be all padding
  double lo[3], vol = 1.0;
  for (int a = 0; a < 3; ++a) {
    lo[a] = (double)root_min[a] - 2.0 * pad;
    ext[a] = ((double)root_max[a] + 2.0 * pad) - lo[a];
    vol *= ext[a];
  }
  // Resolution: about one cell per primitive (density <= 0: a search around that).  What a ray pays is one step per cell
  // it crosses plus one box test per record it comes across, i.e. per unit length ~ r (cells) and ~ refs * r / ncell
  // (records), r = mean resolution; measured on the 10 170-primitive scene a record costs about half a step (kernel
  // time follows refs * r / ncell + 2 r over resolutions 15..31).  The search matters for primitives on a lattice —
  // the resolution that matches the lattice pitch halves the references — and is harmless otherwise.
  auto resolution = [&](double per_len, int res[3]) {
    for (int a = 0; a < 3; ++a) res[a] = (int)std::min(512.0, std::max(1.0, std::floor(ext[a] * per_len + 0.5)));
  };
  auto references = [&](const int res[3]) {
    int64_t refs = 0;
    for (int li : leaves) {
      int64_t n = 1;
      for (int a = 0; a < 3; ++a) {
        int c0 = (int)std::floor(((double)nodes[li].bmin[a] - pad - lo[a]) / ext[a] * res[a]);
        int c1 = (int)std::floor(((double)nodes[li].bmax[a] + pad - lo[a]) / ext[a] * res[a]);
        c0 = std::min(std::max(c0, 0), res[a] - 1);
        c1 = std::min(std::max(c1, 0), res[a] - 1);
        n *= c1 - c0 + 1;
      }
      refs += n;
    }
    return refs;
  };
  const double base = std::cbrt((double)leaves.size() / vol);
  if (density > 0.0) {
    resolution(std::cbrt(density) * base, gb.res);
  } else {
    double best_cost = INFINITY;
    int last[3] = {0, 0, 0};
    for (int i = 0; i <= 40; ++i) {
      int res[3];
      resolution(base * (0.6 + 0.025 * i), res);
      if (res[0] == last[0] && res[1] == last[1] && res[2] == last[2]) continue;
      std::memcpy(last, res, sizeof(last));
      const double r = (res[0] + res[1] + res[2]) / 3.0, cells = (double)res[0] * res[1] * res[2];
      const double cost = (double)references(res) * r / cells + 2.0 * r;
      if (cost < best_cost) best_cost = cost, std::memcpy(gb.res, res, sizeof(res));
    }
  }
  int64_t ncell = 1;
  for (int a = 0; a < 3; ++a) {
    gb.gmin[a] = (float)lo[a];
    gb.cs[a] = (float)(ext[a] / gb.res[a]);
    gb.inv_cs[a] = (float)(gb.res[a] / ext[a]);
    ncell *= gb.res[a];
  }
  gb.pad = (float)pad;
  if (ncell > (int64_t)64 * 1024 * 1024) return false;
  auto cell_range = [&](const ptd::Node& n, int a, int& c0, int& c1) {
    c0 = (int)std::floor(((double)n.bmin[a] - pad - lo[a]) / ext[a] * gb.res[a]);
    c1 = (int)std::floor(((double)n.bmax[a] + pad - lo[a]) / ext[a] * gb.res[a]);
    c0 = std::min(std::max(c0, 0), gb.res[a] - 1);
    c1 = std::min(std::max(c1, 0), gb.res[a] - 1);
  };
  std::vector<uint32_t> count((size_t)ncell + 1, 0);
  int64_t refs = 0;
  for (int li : leaves) {
    int c0[3], c1[3];
    for (int a = 0; a < 3; ++a) cell_range(nodes[li], a, c0[a], c1[a]);
    refs += (int64_t)(c1[0] - c0[0] + 1) * (c1[1] - c0[1] + 1) * (c1[2] - c0[2] + 1);
    if (refs > (int64_t)1 << 22) return false;  // ring entries hold 22-bit record indices
  }
  if (geoms.size() > ((size_t)1 << 24)) return false;  // records hold 24-bit geom indices
  // a candidate only while the lists stay moderate (whether it beats the BVH scan is measured, choose_traversal())
  if (!forced && refs > 64 * (int64_t)leaves.size()) return false;
  for (int li : leaves) {
    int c0[3], c1[3];
    for (int a = 0; a < 3; ++a) cell_range(nodes[li], a, c0[a], c1[a]);
    for (int z = c0[2]; z <= c1[2]; ++z)
      for (int y = c0[1]; y <= c1[1]; ++y)
        for (int x = c0[0]; x <= c1[0]; ++x) ++count[(size_t)x + (size_t)gb.res[0] * ((size_t)y + (size_t)gb.res[1] * z)];
  }
  uint32_t longest = 0;
  gb.start.assign((size_t)ncell + 1, 0);
  for (int64_t c = 0; c < ncell; ++c) {
    gb.start[c + 1] = gb.start[c] + count[c];
    longest = std::max(longest, count[c]);
  }
  if (!forced && longest > 4096) return false;
  gb.items.assign((size_t)refs, ptd::Node{});
  std::vector<uint32_t> fill(gb.start.begin(), gb.start.end() - 1);
  for (int li : leaves) {  // threaded (= reference visiting) order inside every cell
    int c0[3], c1[3];
    for (int a = 0; a < 3; ++a) cell_range(nodes[li], a, c0[a], c1[a]);
    for (int z = c0[2]; z <= c1[2]; ++z)
      for (int y = c0[1]; y <= c1[1]; ++y)
        for (int x = c0[0]; x <= c1[0]; ++x) {
          ptd::Node it = nodes[li];
          it.skip = li;
          it.geom = (x > c0[0] ? 1 : 0) | (x < c1[0] ? 2 : 0) | (y > c0[1] ? 4 : 0) | (y < c1[1] ? 8 : 0) | (z > c0[2] ? 16 : 0) |
                    (z < c1[2] ? 32 : 0) | ((geoms[nodes[li].geom].type & 3) << 6) | (nodes[li].geom << 8);
          gb.items[fill[(size_t)x + (size_t)gb.res[0] * ((size_t)y + (size_t)gb.res[1] * z)]++] = it;
        }
  }
  return true;
}

// A box as centre and half extent for the fast build's slab test (pt_arith.inc slab_t): the centre rounded to float, the half
// extent rounded UP from the distance to the farther face (so the converted box contains the original in real arithmetic), in
// place in (bmin, bmax).  `inner` (inner nodes, subtree entries of the top list — pure acceleration): a little more, so that in the
// test's float arithmetic a ray that passes a leaf's box still passes every box above it (1e-5 of the box and of the
// coordinates: two orders of magnitude above the rounding of the three FMAs).
void center_half_box(float bmin[3], float bmax[3], bool inner) {
  for (int a = 0; a < 3; ++a) {
    const double lo = bmin[a], hi = bmax[a];
    const float c = (float)(0.5 * (lo + hi));
    double h = std::max(hi - (double)c, (double)c - lo);
    if (inner) h = h * (1.0 + 1e-5) + 1e-5 * std::max(std::fabs(lo), std::fabs(hi)) + 1e-30;
    float hf = (float)h;
    if ((double)hf < h) hf = std::nextafter(hf, INFINITY);
    bmin[a] = c, bmax[a] = hf;
  }
}

ptk::SceneTables tables(const Ctx& g) {
  ptk::SceneTables t{};
  t.nodes = g.d_nodes;
  t.num_nodes = g.num_nodes;
  t.geoms = g.d_geoms;
  t.num_geoms = (int)g.geoms.size();
  t.mats = g.d_mats;
  t.num_mats = (int)g.mats.size();
  t.top = g.d_top;
  t.nodes_b = g.d_nodes_b ? g.d_nodes_b : g.d_nodes;
  t.top_b = g.d_top_b ? g.d_top_b : g.d_top;
  t.grid_items_b = nullptr;
  t.num_top = (kAblateBuild && (g.debug_flags & 1)) ? 0 : g.num_top;
  std::memcpy(t.root_min, g.root_min, 12);
  std::memcpy(t.root_max, g.root_max, 12);
  t.cull_margin = (g.debug_flags & 16) ? INFINITY : g.cull_margin;
  t.top_xor = (g.debug_flags & 32) ? 0ull : g.top_xor;
  t.lds_table_bytes = g.lds_table_bytes;
  t.max_batch_iters = g.K;
  t.has_triangles = g.has_triangles ? 1 : 0;
  t.trace_depth = g.depth;
  {
    const char* e = getenv("PT_SCAN_NODES_LDS");  // experiment knob: 0 / 1 = never / always; default: when it costs k_paths no resident workgroup
    t.scan_nodes_lds = e ? (atoi(e) ? 1 : 0) : -1;
  }
  // debug_flags 256 builds and uses the grid for any scene, 512 never (A/B, same results)
  t.use_grid = g.have_grid && g.grid_enabled && !(g.debug_flags & 512) ? 1 : 0;
  if (t.use_grid) {
    t.lds_table_bytes = -1;  // (a forced grid on a small scene: the grid kernels read the tables from memory)
    t.grid_start = g.d_grid_start + g.grid_guard;
    t.grid_items = g.d_grid_items;
    t.grid_items_b = g.d_grid_items_b ? g.d_grid_items_b : g.d_grid_items;
    for (int a = 0; a < 3; ++a)
      t.grid_res[a] = g.grid_res[a], t.grid_min[a] = g.grid_min[a], t.grid_cs[a] = g.grid_cs[a], t.grid_inv_cs[a] = g.grid_inv_cs[a];
    t.grid_pad = g.grid_pad;
  }
  return t;
}

int alloc_pathbuf(Ctx& g, ptd::PathBuf* b, int64_t stride) {
  b->stride = stride;
  // planes 0 and 1: 16 bytes per path, plane 2: kPathPlane2Bytes (pt_device.h); allocated in 16-byte words
  return dalloc(g, &b->r, 2 * stride + (stride * ptd::kPathPlane2Bytes + 15) / 16);
}
int alloc_hitbuf(Ctx& g, ptd::HitBuf* h, int64_t stride) {
  h->stride = stride;
  if (dalloc(g, &h->t, stride)) return -1;
  if (dalloc(g, &h->n, 3 * stride)) return -1;
  if (dalloc(g, &h->mat, stride)) return -1;
  if (dalloc(g, &h->p, 3 * stride)) return -1;
  return 0;
}

// Queue descriptor for a launch of `grid` workgroups (W = waves of THAT launch; Q, cap shared).
ptd::Queues queues_for(const Ctx& g, int grid) {
  ptd::Queues q = g.qs;
  q.W = grid * ptk::kWavesPerBlock;
  return q;
}

int run_batch(Ctx& g, int iter_first, int kb) {
  ptk::BatchInfo b{};
  b.iter_first = iter_first;
  b.K = kb;
  b.N = g.N;
  b.pixel_begin = g.pixel_begin;
  b.trace_depth = g.depth;
  b.slot_shift = g.slot_shift;
  b.aa_jitter = g.aa_jitter ? 1 : 0;
  b.debug = kAblateBuild ? g.debug_flags : 0;
  b.flat = g.fuse_bounces ? 0 : 1;
  {
    // k_primary's strands in pieces (BatchInfo::primary_pieces): a piece pays for its own pipeline drain, so it should hold a
    // few dozen 64-sample groups; a wave's strand has K * nq / (waves per queue) of them
    const char* e = getenv("PT_PRIMARY_PIECES");  // experiment knob
    const int64_t groups = (int64_t)kb * g.qs.nq / std::max(1, g.ret.wq0);
    b.primary_pieces = e ? atoi(e) : (int)std::min<int64_t>(4, std::max<int64_t>(1, groups / 48));
    const char* ep = getenv("PT_PATHS_PIECES");  // experiment knob
    const char* em = getenv("PT_PATHS_MIN_PIECE");  // test knob: pieces of a few paths, so that small images exercise the piece switches
    b.paths_pieces = (ep ? std::min(std::max(atoi(ep), 1), 0x7fff) : 2) | (em ? std::min(std::max(atoi(em), 1), 0x7fff) : 64) << 16;
  }
  b.stripe = g.stripe;
  b.gap = g.stripe ? g.stripe_stride - g.stripe : 0;
  b.inv_stripe = g.stripe ? 1.0f / (float)g.stripe : 0.0f;
  const ptk::SceneTables sc = tables(g);
  const ptk::KernelApi& k = *g.k;
  const size_t per_depth = (size_t)g.qs.Q * g.qs.cnt_stride;
  int d0 = 0;
  if (g.fuse_primary) {
    // depth 0 in one launch; its survivors are the depth-1 input (buf[1], cnt[1])
    k.primary(g.stream, g.grid_primary, sc, g.dcam, b, queues_for(g, g.grid_primary), g.d_cnt, g.d_cnt + per_depth, g.buf[1],
              g.ret);
    d0 = 1;
  } else {
    k.generate(g.stream, g.grid_gen, g.dcam, b, queues_for(g, g.grid_gen), g.buf[0], g.d_cnt);
  }
  int src = d0 & 1;
  // Depths >= 1, fused: ONE launch of k_paths — persistent lanes with their own depth, no path state through HBM after depth 0
  const bool all_depths = g.grid_paths > 0;
  if (all_depths) {
    EventPair ev{};
    if (g.time_kernels) {
      if (get_events(g, &ev)) return -1;
      HIP_OK(hipEventRecord(ev.a, g.stream));
    }
    k.paths(g.stream, g.grid_paths, sc, b, queues_for(g, g.grid_paths), g.d_cnt, g.buf[1], g.ret);
    if (g.time_kernels) {
      HIP_OK(hipEventRecord(ev.b, g.stream));
      g.pending_isect.push_back(ev);
    }
  }
  // ... unfused (tests, A/B): computeIntersections and shadeAndExtendRays as separate launches per depth
  for (int d = d0; d < g.depth && !all_depths; ++d) {
    const int32_t* cin = g.d_cnt + per_depth * d;
    int32_t* cout = g.d_cnt + per_depth * (d + 1);
    EventPair ev{};
    if (g.time_kernels) {  // brackets the dominant kernel of this depth
      if (get_events(g, &ev)) return -1;
      HIP_OK(hipEventRecord(ev.a, g.stream));
    }
    k.intersect(g.stream, g.grid_isect, sc, queues_for(g, g.grid_isect), cin, g.buf[src], g.hits, g.legacy, d == 0);
    if (g.time_kernels) {
      HIP_OK(hipEventRecord(ev.b, g.stream));
      g.pending_isect.push_back(ev);
    }
    k.shade(g.stream, g.grid_shade, sc, b, d, queues_for(g, g.grid_shade), cin, cout, g.buf[src], g.hits, g.buf[src ^ 1], g.ret);
    src ^= 1;
  }
  if (getenv("PT_DUMP_QUEUE_BALANCE")) {  // diagnostics: fill levels of the queues per depth (before k_count_stats zeroes them)
    HIP_OK(hipStreamSynchronize(g.stream));
    std::vector<int32_t> h((size_t)(g.depth + 1) * per_depth);
    HIP_OK(hipMemcpy(h.data(), g.d_cnt, h.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    for (int d = 0; d <= g.depth; ++d) {
      double sum = 0, mx = 0, mn = 1e30;
      for (int q = 0; q < g.qs.Q; ++q) {
        const double v = h[(size_t)d * per_depth + (size_t)q * g.qs.cnt_stride];
        sum += v, mx = std::max(mx, v), mn = std::min(mn, v);
      }
      std::fprintf(stderr, "queue balance depth %d: mean %.0f min %.0f max %.0f (max/mean %.3f)\n", d, sum / g.qs.Q, mn, mx, sum > 0 ? mx / (sum / g.qs.Q) : 0.0);
    }
  }
  k.count_stats(g.stream, g.qs, g.d_cnt, g.depth, g.d_stats);
  k.collect(g.stream, b, g.qs, g.ret, g.d_image);
  HIP_OK(hipGetLastError());
  g.samples += (int64_t)kb * g.N;
  if (g.pending_isect.size() > 16384) {
    HIP_OK(hipStreamSynchronize(g.stream));
    if (resolve_events(g)) return -1;
  }
  return 0;
}

// Launch geometry (persistent grids = resident workgroups) for the kernels the current tables select.
void plan_launch(Ctx& g) {
  const char* kb = getenv("PT_LDS_TABLE_KB");  // test / experiment knob: force the LDS staging limit of the scene tables
  g.lds_table_bytes = g.k->lds_table_limit(tables(g), kb ? atoi(kb) * 1024 : -1);
  const ptk::SceneTables t = tables(g);
  g.grid_gen = g.num_cus * std::min(g.cap_bpc, g.k->resident_blocks_per_cu(ptk::kGenerate, t));
  g.grid_isect = g.num_cus * std::min(g.cap_bpc, g.k->resident_blocks_per_cu(g.legacy ? ptk::kIntersectLegacy : ptk::kIntersect, t));
  g.grid_shade = g.num_cus * std::min(g.cap_bpc, g.k->resident_blocks_per_cu(ptk::kShade, t));
  g.grid_primary = g.num_cus * std::min(g.cap_bpc, g.k->resident_blocks_per_cu(ptk::kPrimary, t));
  g.ret.wq0 = g.grid_primary * ptk::kWavesPerBlock / g.qs.Q;  // the sub-lists' residue count (pt_device.h RetireBuf)
  g.grid_paths = (g.fuse_bounces && g.depth > 1) ? g.num_cus * std::min(g.cap_bpc, g.k->resident_blocks_per_cu(ptk::kPaths, t)) : 0;
  g.qs.paths_W = g.grid_paths * ptk::kWavesPerBlock;  // what k_count_stats deals to the queues (a table made for another width is ignored)
}

// Make candidate i the grid the kernels walk.
void use_grid_alt(Ctx& g, size_t i) {
  const Ctx::GridAlt& a = g.grid_alts[i];
  g.d_grid_start = a.d_start, g.grid_guard = a.guard, g.d_grid_items = a.d_items, g.d_grid_items_b = a.d_items_b;
  for (int k = 0; k < 3; ++k) g.grid_res[k] = a.res[k], g.grid_min[k] = a.gmin[k], g.grid_cs[k] = a.cs[k], g.grid_inv_cs[k] = a.inv_cs[k];
  g.grid_pad = a.pad;
  g.have_grid = true;
}
// Grid or BVH scan, and which grid?  All give the same image (LABNOTES.md section 9.1); which is faster depends on how the
// primitives are spread and how far rays fly, and no count of references predicted it across lattice, random and clustered
// scenes — so it is measured: up to 8 iterations of the context's own tile with each (the second of two runs counts), before
// the first sample is rendered.  Costs a few tens of milliseconds per candidate for a 1080p tile.  debug_flags 256 / 512 skip
// the measurement.
int choose_traversal(Ctx& g) {
  auto drop_unused_grids = [&](size_t keep) {
    for (size_t i = 0; i < g.grid_alts.size(); ++i) {
      if (i == keep) continue;
      for (void* p : {(void*)g.grid_alts[i].d_start, (void*)g.grid_alts[i].d_items, g.grid_alts[i].d_items_b != g.grid_alts[i].d_items ? (void*)g.grid_alts[i].d_items_b : nullptr}) {
        if (!p) continue;
        (void)hipFree(p);
        g.allocs.erase(std::remove(g.allocs.begin(), g.allocs.end(), p), g.allocs.end());
      }
      g.device_bytes -= (int64_t)g.grid_alts[i].bytes;
    }
    g.grid_alts.clear();
  };
  if (!g.have_grid || !g.fuse_bounces || (g.debug_flags & (256 | 512))) {
    if (g.have_grid) drop_unused_grids(0);
    return 0;
  }
  auto time_one = [&](float* ms) -> int {
    plan_launch(g);
    for (int rep = 0; rep < 2; ++rep) {
      EventPair ev{};
      if (get_events(g, &ev)) return -1;
      HIP_OK(hipEventRecord(ev.a, g.stream));
      if (run_batch(g, 1, std::min(g.K, 8))) return -1;  // enough groups per wave for the persistent grids to fill
      HIP_OK(hipEventRecord(ev.b, g.stream));
      HIP_OK(hipStreamSynchronize(g.stream));
      HIP_OK(hipEventElapsedTime(ms, ev.a, ev.b));
      g.free_events.push_back(ev);
    }
    return 0;
  };
  g.grid_enabled = false;
  if (time_one(&g.probe_ms[0])) return -1;
  g.grid_enabled = true;
  size_t best = 0;
  for (size_t i = 0; i < g.grid_alts.size(); ++i) {
    float ms = 0.f;
    use_grid_alt(g, i);
    if (time_one(&ms)) return -1;
    // a finer grid has to beat the model's by 3 %, so that the choice does not flip between runs on timing noise
    if (i == 0 || ms < 0.97f * g.probe_ms[1]) g.probe_ms[1] = ms, best = i;
  }
  use_grid_alt(g, best);
  drop_unused_grids(best);
  // the grid has to win by a margin: two short timed runs on a shared or noisy box differ by a few per cent, and a
  // choice that flips between runs makes PtStats and profiles irreproducible; results are equal either way
  g.grid_enabled = g.probe_ms[1] < 0.95f * g.probe_ms[0];
  plan_launch(g);
  HIP_OK(hipMemsetAsync(g.d_image, 0, 3 * (size_t)g.N * sizeof(float), g.stream));
  HIP_OK(hipMemsetAsync(g.d_stats, 0, PT_MAX_DEPTH * sizeof(unsigned long long), g.stream));
  HIP_OK(hipStreamSynchronize(g.stream));
  g.samples = 0;
  return 0;
}

struct Scratch {  // frees on scope exit
  std::vector<void*> p;
  ~Scratch() {
    for (void* q : p) (void)hipFree(q);
  }
  template <typename T>
  T* get(size_t n) {
    void* q = nullptr;
    if (hipMalloc(&q, std::max<size_t>(n * sizeof(T), 16)) != hipSuccess) return nullptr;
    p.push_back(q);
    return reinterpret_cast<T*>(q);
  }
};
ptd::Queues single_queue(const Ctx& g, int n) {
  ptd::Queues qs{};
  qs.Q = 1;
  qs.cap = ((n + 63) / 64) * 64;
  qs.W = g.grid * ptk::kWavesPerBlock;
  qs.cnt_stride = 16;
  qs.nq = (n + 63) / 64;
  qs.inv_nq = 1.0f / (float)qs.nq;
  return qs;
}

void destroy(Ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  for (void* p : c->allocs) (void)hipFree(p);
  auto kill = [](std::vector<EventPair>& v) {
    for (auto& e : v) {
      (void)hipEventDestroy(e.a);
      (void)hipEventDestroy(e.b);
    }
    v.clear();
  };
  kill(c->free_events);
  kill(c->pending_isect);
  kill(c->pending_render);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

// pathtraceInit (pathtrace.cu:462-516) for one context.  Any failure leaves nothing behind: the caller destroys `g`.
int setup(Ctx& g, const PtSceneDesc* sc, const PtOptions& opt) {
  const int W = sc->camera.resolution[0], H = sc->camera.resolution[1];
  g.arith = opt.arith;
  g.k = ptk::api_for(opt.arith);
  if (!g.k) return fail("pt_init: arith %d is not one of PT_ARITH_EXACT / PT_ARITH_FMA / PT_ARITH_FAST", opt.arith);
  if (!kAblateBuild && (opt.debug_flags & 15))
    return fail("pt_init: debug_flags bits 0-3 (ablations with wrong results) exist only in -DPT_ABLATE builds of the library");
  int ndev = 0;
  HIP_OK(hipGetDeviceCount(&ndev));
  if (ndev <= 0) return fail("pt_init: no HIP device (this library has no CPU fallback)");
  if (opt.device < 0 || opt.device >= ndev) return fail("pt_init: device %d of %d", opt.device, ndev);
  g.device = opt.device;
  HIP_OK(hipSetDevice(g.device));
  hipDeviceProp_t prop;
  HIP_OK(hipGetDeviceProperties(&prop, g.device));
  g.num_cus = prop.multiProcessorCount;
  HIP_OK(hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking));

  g.geoms.assign(sc->geoms, sc->geoms + sc->num_geoms);
  g.mats.assign(sc->materials, sc->materials + sc->num_materials);
  g.cam = sc->camera;
  g.depth = sc->trace_depth;
  g.dcam.res_x = W, g.dcam.res_y = H;
  std::memcpy(g.dcam.pos, g.cam.position, 12);
  std::memcpy(g.dcam.view, g.cam.view, 12);
  std::memcpy(g.dcam.up, g.cam.up, 12);
  std::memcpy(g.dcam.right, g.cam.right, 12);
  g.dcam.pl_x = g.cam.pixelLength[0], g.dcam.pl_y = g.cam.pixelLength[1];

  g.pixel_begin = opt.pixel_begin;
  g.N = opt.pixel_count > 0 ? opt.pixel_count : W * H - opt.pixel_begin;
  g.stripe = opt.stripe_pixels;
  g.stripe_stride = opt.stripe_stride;
  if (g.stripe < 0 || (g.stripe > 0 && g.stripe_stride < g.stripe)) return fail("pt_init: bad stripe %d / stride %d", g.stripe, g.stripe_stride);
  {
    // last global pixel the tile touches
    int64_t last = (int64_t)g.pixel_begin + g.N - 1;
    if (g.stripe > 0) last = (int64_t)g.pixel_begin + (g.N - 1) + (int64_t)((g.N - 1) / g.stripe) * (g.stripe_stride - g.stripe);
    if (g.pixel_begin < 0 || g.N <= 0 || last >= (int64_t)W * H)
      return fail("pt_init: tile [%d, +%d, stripe %d/%d) outside %dx%d", opt.pixel_begin, opt.pixel_count, g.stripe, g.stripe_stride, W, H);
    if (g.stripe > 0 && g.N / g.stripe >= 32768) return fail("pt_init: more than 32767 stripes");
    if (H >= 32768) return fail("pt_init: image height %d not supported (>= 32768)", H);
  }
  // batch size: enough paths in flight to fill the chip a few times over; slots are int32
  int K = opt.iters_per_batch;
  if (K <= 0) {
    const int64_t target = 48ll << 20;  // ~50 M paths per batch (24 iterations of 1920x1080, ~5 GB of path state):
                                        // launch boundaries drop below 1 % of a batch (measured K=6 → 48: +8 %)
    K = (int)std::max<int64_t>(1, std::min<int64_t>(256, (target + g.N - 1) / g.N));
  }
  // sample ids are k << slot_shift | tile pixel in 31 bits (BatchInfo::slot_shift)
  g.slot_shift = 1;
  while ((1ll << g.slot_shift) < g.N) ++g.slot_shift;
  if (g.slot_shift > 30) return fail("pt_init: tiles of more than 2^30 pixels are not supported");
  while (K > 1 && K > (1 << (31 - g.slot_shift))) --K;

  // Compaction queues: every launch needs (waves % Q) == 0, so Q divides 4 * CUs.  A queue owns every Q-th 64-pixel
  // chunk of the tile in every iteration (pt_device.h); k_collect wants at most 128 chunks per queue (one LDS tile per
  // pass), so large tiles get more queues while the waves allow it.
  int Q = opt.num_queues > 0 ? opt.num_queues : 256;
  const int cu_waves = g.num_cus * ptk::kWavesPerBlock;
  if (Q > cu_waves) Q = cu_waves;
  while (cu_waves % Q) --Q;
  const int64_t chunks = ((int64_t)g.N + 63) / 64;  // per iteration
  if (opt.num_queues <= 0)
    while ((chunks + Q - 1) / Q > 128 && 2 * Q <= cu_waves && cu_waves % (2 * Q) == 0) Q *= 2;
  const int nq = (int)((chunks + Q - 1) / Q);
  const int grid = g.num_cus * 8;
  // Retirement records: one slot per (iteration of the batch, pixel) — regions (queue, iteration) of nq * 64 slots, exactly
  // full at the end of a batch (pt_device.h RetireBuf): 16 bytes per sample and batch.
  g.K = K;
  g.grid = grid;
  g.qs.Q = Q;
  g.qs.W = grid * ptk::kWavesPerBlock;
  g.qs.cnt_stride = 16;
  g.qs.nq = nq;
  g.qs.inv_nq = 1.0f / (float)nq;
  g.qs.cap = K * nq * 64;
  g.stride = (int64_t)Q * g.qs.cap;
  g.ret.seg_cap = nq * 64;
  g.ret.kmax = K;

  // scene tables
  std::vector<PtBVHNode> ref_nodes;
  pt::buildBVH(g.geoms.data(), (int)g.geoms.size(), ref_nodes);
  std::vector<ptd::Node> nodes;
  nodes.reserve(ref_nodes.size());
  std::vector<int> where(ref_nodes.size(), -1);
  thread_bvh(ref_nodes, 0, nodes, where);
  g.num_nodes = (int)nodes.size();
  std::vector<ptd::TopEntry> top;
  build_top(ref_nodes, nodes, where, g.geoms, top, &g.top_xor);
  g.num_top = (int)top.size();
  std::memcpy(g.root_min, ref_nodes[0].bmin, 12);
  std::memcpy(g.root_max, ref_nodes[0].bmax, 12);
  g.tight_leaves = 0;
  if (g.num_nodes >= kTightNodes && !(opt.debug_flags & 2048)) {
    double olo[3], ohi[3];
    origin_region(g.root_min, g.root_max, g.cam.position, olo, ohi);
    g.tight_leaves = tighten_sphere_leaves(nodes, g.geoms, olo, ohi);
    for (ptd::TopEntry& e : top)
      if (e.link < 0) std::memcpy(e.bmin, nodes[e.idx].bmin, 12), std::memcpy(e.bmax, nodes[e.idx].bmax, 12);
  }
  {
    // SceneTables::cull_margin: >= 10x the worst undershoot of a reported hit distance (1e-4 object units mapped
    // to world space + rounding of the transforms at the scene's coordinate magnitudes)
    float max_xf = 1.0f, extent = 1.0f;
    for (const PtGeom& gm : g.geoms) {
      if (gm.type == PT_GEOM_TRIANGLE) continue;  // world-space test: the pull-back is 1e-4 world units (max_xf >= 1 covers it)
      float f = 0.f;
      for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) f += gm.transform[c * 4 + r] * gm.transform[c * 4 + r];
      max_xf = std::max(max_xf, sqrtf(f));
    }
    for (int a = 0; a < 3; ++a)
      extent = std::max({extent, fabsf(g.root_min[a]), fabsf(g.root_max[a]), fabsf(g.cam.position[a])});
    g.cull_margin = 1e-3f * max_xf + 1e-4f * extent;
  }
  g.legacy = opt.legacy_traversal != 0;
  g.aa_jitter = opt.aa_jitter != 0;
  g.debug_flags = opt.debug_flags;
  g.fuse_primary = !g.legacy && !opt.unfused_primary;
  g.fuse_bounces = g.fuse_primary && !opt.unfused_bounces;
  std::vector<ptd::Geom> dg(g.geoms.size());
  for (size_t i = 0; i < g.geoms.size(); ++i) {
    std::memset(&dg[i], 0, sizeof(ptd::Geom));
    dg[i].type = g.geoms[i].type;
    dg[i].material = g.geoms[i].materialid;
    if (g.geoms[i].type == PT_GEOM_TRIANGLE) {  // mesh extension: the three world-space vertices where `inv` would be
      std::memcpy(dg[i].inv, g.geoms[i].transform, 9 * sizeof(float));
      g.has_triangles = true;
      continue;
    }
    pack_rows(g.geoms[i].inverseTransform, dg[i].inv);
    pack_rows(g.geoms[i].transform, dg[i].xf);
    pack_rows(g.geoms[i].invTranspose, dg[i].invT);
    box_normal_table(dg[i].invT, dg[i].box_normal);
  }
  std::vector<ptd::Mat> dm(g.mats.size());
  for (size_t i = 0; i < g.mats.size(); ++i) {
    std::memset(&dm[i], 0, sizeof(ptd::Mat));
    std::memcpy(dm[i].color, g.mats[i].color, 12);
    std::memcpy(dm[i].spec, g.mats[i].specular_color, 12);
    dm[i].reflective = g.mats[i].hasReflective;
    dm[i].refractive = g.mats[i].hasRefractive;
    dm[i].emittance = g.mats[i].emittance;
  }
  if (dm.size() * sizeof(ptd::Mat) > 60 * 1024) return fail("pt_init: %zu materials exceed the LDS table", dm.size());
  if (dalloc(g, &g.d_nodes, nodes.size()) || dalloc(g, &g.d_geoms, dg.size()) || dalloc(g, &g.d_mats, dm.size()) ||
      dalloc(g, &g.d_top, top.size()))
    return -1;
  HIP_OK(hipMemcpy(g.d_top, top.data(), top.size() * sizeof(ptd::TopEntry), hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(g.d_nodes, nodes.data(), nodes.size() * sizeof(ptd::Node), hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(g.d_geoms, dg.data(), dg.size() * sizeof(ptd::Geom), hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(g.d_mats, dm.data(), dm.size() * sizeof(ptd::Mat), hipMemcpyHostToDevice));
  if (g.k->boxes_center_half) {  // the fast build: converted copies for the bounce kernels
    std::vector<ptd::Node> nb = nodes;
    for (ptd::Node& n : nb) center_half_box(n.bmin, n.bmax, n.geom < 0);
    std::vector<ptd::TopEntry> tb = top;
    for (ptd::TopEntry& e : tb) center_half_box(e.bmin, e.bmax, e.link >= 0);
    if (dalloc(g, &g.d_nodes_b, nb.size()) || dalloc(g, &g.d_top_b, tb.size())) return -1;
    HIP_OK(hipMemcpy(g.d_nodes_b, nb.data(), nb.size() * sizeof(ptd::Node), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(g.d_top_b, tb.data(), tb.size() * sizeof(ptd::TopEntry), hipMemcpyHostToDevice));
  }
  if ((g.num_nodes >= kGridNodes || (g.debug_flags & 256)) && !(g.debug_flags & 512)) {
    const char* dens = getenv("PT_GRID_DENSITY");  // experiment knob: cells per primitive instead of the search
    double cam_mag = 0.0;
    for (int a = 0; a < 3; ++a) cam_mag = std::max(cam_mag, std::fabs((double)g.cam.position[a]));
    // Candidates: the resolution the host's cost model finds (about one cell per primitive; build_grid) and two finer ones.
    // The model prices a uniform spread of primitives and rays; where the primitives cluster, a finer grid pays (clustered 1500
    // objects: 2319 Msamples/s at the model's resolution, 2931 at 8 cells per primitive; a lattice wants exactly its pitch),
    // so choose_traversal() measures.  A forced grid (debug_flags 256) or a forced density is the only candidate.
    const bool forced = (g.debug_flags & 256) != 0;
    std::vector<double> densities{dens ? atof(dens) : 0.0};
    if (!forced && !dens) densities.insert(densities.end(), {4.0, 8.0});
    for (double density : densities) {
      GridBuild gb;
      if (!build_grid(nodes, g.geoms, g.root_min, g.root_max, cam_mag, density, forced, gb)) continue;
      bool repeat = false;
      for (const Ctx::GridAlt& o : g.grid_alts) repeat = repeat || (o.res[0] == gb.res[0] && o.res[1] == gb.res[1] && o.res[2] == gb.res[2]);
      if (repeat) continue;
      Ctx::GridAlt alt{};
      // the cell table with empty cells before and after it: a walk may run up to one cell per axis past the grid's far side
      // before its distance test ends it (pt_kernels.hip CellWalk)
      alt.guard = (size_t)gb.res[0] * gb.res[1] + gb.res[0] + 2;
      std::vector<uint32_t> padded(gb.start.size() + 2 * alt.guard, 0u);
      std::copy(gb.start.begin(), gb.start.end(), padded.begin() + alt.guard);
      std::fill(padded.begin() + alt.guard + gb.start.size(), padded.end(), gb.start.back());
      const int64_t before = g.device_bytes;
      if (dalloc(g, &alt.d_start, padded.size()) || dalloc(g, &alt.d_items, gb.items.size())) return -1;
      HIP_OK(hipMemcpy(alt.d_start, padded.data(), padded.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
      HIP_OK(hipMemcpy(alt.d_items, gb.items.data(), gb.items.size() * sizeof(ptd::Node), hipMemcpyHostToDevice));
      alt.d_items_b = alt.d_items;
      if (g.k->boxes_center_half) {  // the grid's records are leaf boxes
        std::vector<ptd::Node> ib = gb.items;
        for (ptd::Node& n : ib) center_half_box(n.bmin, n.bmax, false);
        if (dalloc(g, &alt.d_items_b, ib.size())) return -1;
        HIP_OK(hipMemcpy(alt.d_items_b, ib.data(), ib.size() * sizeof(ptd::Node), hipMemcpyHostToDevice));
      }
      alt.bytes = (size_t)(g.device_bytes - before);
      for (int a = 0; a < 3; ++a) alt.res[a] = gb.res[a], alt.gmin[a] = gb.gmin[a], alt.cs[a] = gb.cs[a], alt.inv_cs[a] = gb.inv_cs[a];
      alt.pad = gb.pad;
      g.grid_alts.push_back(alt);
    }
    if (!g.grid_alts.empty()) use_grid_alt(g, 0);
  }

  g.cap_bpc = opt.blocks_per_cu > 0 ? std::min(opt.blocks_per_cu, 8) : 8;
  plan_launch(g);
  // path state
  if (alloc_pathbuf(g, &g.buf[0], g.stride) || alloc_pathbuf(g, &g.buf[1], g.stride)) return -1;
  if (!g.fuse_bounces && alloc_hitbuf(g, &g.hits, g.stride)) return -1;  // hit records reach HBM only in the unfused form
  {
    const size_t regions = (size_t)Q * g.ret.kmax;
    const size_t wq_max = (size_t)g.num_cus * 8 * ptk::kWavesPerBlock / Q;  // waves per queue of the widest k_primary grid
    if (dalloc(g, &g.ret.rec, regions * g.ret.seg_cap) || dalloc(g, &g.ret.cnt, regions) || dalloc(g, &g.ret.sub, regions * wq_max)) return -1;
    HIP_OK(hipMemset(g.ret.cnt, 0, regions * sizeof(unsigned long long)));  // k_collect re-zeroes them after every batch
    HIP_OK(hipMemset(g.ret.sub, 0, regions * wq_max * sizeof(unsigned long long)));
  }
  if (dalloc(g, &g.d_image, 3 * (size_t)g.N)) return -1;
  if (dalloc(g, &g.d_cnt, (size_t)(g.depth + 1) * Q * g.qs.cnt_stride)) return -1;
  HIP_OK(hipMemset(g.d_cnt, 0, (size_t)(g.depth + 1) * Q * g.qs.cnt_stride * sizeof(int32_t)));  // k_count_stats re-zeroes it after every batch
  if (dalloc(g, &g.d_stats, PT_MAX_DEPTH)) return -1;
  if (!getenv("PT_NO_DEAL")) {  // (A/B knob: W / Q waves per queue in every batch; same image)
    if (dalloc(g, &g.qs.deal, 4 * (size_t)Q + 2)) return -1;
    HIP_OK(hipMemset(g.qs.deal, 0, (4 * (size_t)Q + 2) * sizeof(int32_t)));  // nothing measured yet: W / Q each
  }
  HIP_OK(hipMemset(g.d_image, 0, 3 * (size_t)g.N * sizeof(float)));
  HIP_OK(hipMemset(g.d_stats, 0, PT_MAX_DEPTH * sizeof(unsigned long long)));
  HIP_OK(hipDeviceSynchronize());
  if (choose_traversal(g)) return -1;
  g.time_kernels = opt.time_kernels != 0;
  return 0;
}

int need(const PtContext* c, const char* who) {
  if (!c) return fail("%s: pt_init has not been called", who);
  return 0;
}

}  // namespace

extern "C" {

const char* pt_last_error(void) { return g_err.c_str(); }
int pt_library_has_ablations(void) { return kAblateBuild ? 1 : 0; }

int pt_selfcheck_ieee(int arith, int kind, uint64_t first, uint64_t count, uint32_t seed, uint64_t* mismatches) {
  const ptk::KernelApi* k = ptk::api_for(arith);
  if (!k || kind < 0 || kind > 4 || !mismatches) return fail("pt_selfcheck_ieee: bad argument");
  unsigned long long* d = nullptr;
  HIP_OK(hipMalloc(&d, sizeof(*d)));
  hipError_t e = hipMemset(d, 0, sizeof(*d));
  if (e == hipSuccess) {
    k->ieee_check(nullptr, kind, first, count, seed, d);
    e = hipGetLastError();
  }
  unsigned long long h = 0;
  if (e == hipSuccess) e = hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) return fail("pt_selfcheck_ieee: %s", hipGetErrorString(e));
  *mismatches = h;
  return 0;
}

// ---- scene -----------------------------------------------------------------
struct PtScene {
  pt::Scene scene;
  explicit PtScene(const std::string& f) : scene(f) {}
};

int pt_scene_load(const char* path, int res_w, int res_h, PtScene** out) {
  if (!path || !out) return fail("pt_scene_load: null argument");
  try {
    PtScene* s = new PtScene(path);
    if (res_w > 0 && res_h > 0) s->scene.overrideResolution(res_w, res_h);
    s->scene.applyInitialCameraState();
    *out = s;
    return 0;
  } catch (const std::exception& e) {
    return fail("pt_scene_load: %s", e.what());
  }
}
void pt_scene_free(PtScene* s) { delete s; }
int pt_scene_desc(const PtScene* s, PtSceneDesc* out) {
  if (!s || !out) return fail("pt_scene_desc: null argument");
  *out = s->scene.desc();
  return 0;
}
int pt_scene_iterations(const PtScene* s) { return s ? (int)s->scene.state.iterations : 0; }
const char* pt_scene_image_name(const PtScene* s) { return s ? s->scene.state.imageName.c_str() : ""; }

int pt_build_bvh(const PtGeom* geoms, int num_geoms, PtBVHNode* out, int cap) {
  std::vector<PtBVHNode> nodes;
  pt::buildBVH(geoms, num_geoms, nodes);
  if (out) std::memcpy(out, nodes.data(), sizeof(PtBVHNode) * std::min<size_t>(nodes.size(), (size_t)std::max(cap, 0)));
  return (int)nodes.size();
}

int pt_traversal_boxes(const PtGeom* geoms, int num_geoms, const float camera_position[3], float* boxes) {
  if (!geoms || num_geoms <= 0 || !camera_position || !boxes) return fail("pt_traversal_boxes: null argument");
  std::vector<PtBVHNode> ref_nodes;
  pt::buildBVH(geoms, num_geoms, ref_nodes);
  double olo[3], ohi[3];
  origin_region(ref_nodes[0].bmin, ref_nodes[0].bmax, camera_position, olo, ohi);
  int tightened = 0;
  for (const PtBVHNode& n : ref_nodes) {
    if (n.left >= 0) continue;
    float* b = boxes + 6 * (size_t)n.geomIndex;
    std::memcpy(b, n.bmin, 12);
    std::memcpy(b + 3, n.bmax, 12);
    float lo[3], hi[3];
    if (geoms[n.geomIndex].type == PT_GEOM_SPHERE && sphere_tight_box(geoms[n.geomIndex], olo, ohi, n.bmin, n.bmax, lo, hi)) {
      std::memcpy(b, lo, 12);
      std::memcpy(b + 3, hi, 12);
      ++tightened;
    }
  }
  return tightened;
}

int pt_build_grid(const PtGeom* geoms, int num_geoms, int forced, PtGridInfo* info, uint32_t* cell_start, PtGridRecord* records) {
  if (!geoms || num_geoms <= 0 || !info) return fail("pt_build_grid: null argument");
  std::vector<PtBVHNode> ref_nodes;
  pt::buildBVH(geoms, num_geoms, ref_nodes);
  std::vector<ptd::Node> nodes;
  std::vector<int> where(ref_nodes.size(), -1);
  thread_bvh(ref_nodes, 0, nodes, where);
  std::memset(info, 0, sizeof(*info));
  info->num_leaves = num_geoms;
  if (!forced && (int)nodes.size() < kGridNodes) return 0;
  GridBuild gb;
  const char* dens = getenv("PT_GRID_DENSITY");
  if (!build_grid(nodes, std::vector<PtGeom>(geoms, geoms + num_geoms), ref_nodes[0].bmin, ref_nodes[0].bmax, 0.0, dens ? atof(dens) : 0.0, forced != 0, gb)) return 0;
  for (int a = 0; a < 3; ++a) info->res[a] = gb.res[a], info->origin[a] = gb.gmin[a], info->cell_size[a] = gb.cs[a];
  info->pad = gb.pad;
  info->num_cells = (int32_t)(gb.start.size() - 1);
  info->num_records = (int32_t)gb.items.size();
  if (cell_start) std::memcpy(cell_start, gb.start.data(), gb.start.size() * sizeof(uint32_t));
  static_assert(sizeof(PtGridRecord) == sizeof(ptd::Node), "PtGridRecord mirrors the device record");
  if (records) std::memcpy(records, gb.items.data(), gb.items.size() * sizeof(ptd::Node));
  return 1;
}

void pt_center_half_box(const float lo[3], const float hi[3], int inner, float center[3], float half_extent[3]) {
  float a[3] = {lo[0], lo[1], lo[2]}, b[3] = {hi[0], hi[1], hi[2]};
  center_half_box(a, b, inner != 0);
  std::memcpy(center, a, 12);
  std::memcpy(half_extent, b, 12);
}

int pt_build_transform(const float* trs, float* transform, float* inverse, float* invTranspose) {
  if (!trs || !transform || !inverse || !invTranspose) return fail("pt_build_transform: null argument");
  pt::buildTransform(trs, transform, inverse, invTranspose);
  return 0;
}

// ---- renderer contexts ---------------------------------------------------------
int pt_ctx_create(const PtSceneDesc* sc, const PtOptions* opt_in, PtContext** out) {
  if (!out) return fail("pt_ctx_create: null output");
  *out = nullptr;
  if (!sc) return fail("pt_init: null scene");
  if (sc->num_geoms <= 0 || !sc->geoms) return fail("pt_init: scene has no geometry");
  if (sc->num_materials <= 0 || !sc->materials) return fail("pt_init: scene has no materials");
  if (sc->trace_depth <= 0 || sc->trace_depth > PT_MAX_DEPTH) return fail("pt_init: trace_depth %d out of range", sc->trace_depth);
  const int W = sc->camera.resolution[0], H = sc->camera.resolution[1];
  if (W <= 0 || H <= 0 || (int64_t)W * H > (1ll << 30)) return fail("pt_init: bad resolution %dx%d", W, H);
  for (int i = 0; i < sc->num_geoms; ++i) {
    if (sc->geoms[i].materialid < 0 || sc->geoms[i].materialid >= sc->num_materials)
      return fail("pt_init: geom %d references material %d of %d", i, sc->geoms[i].materialid, sc->num_materials);
    if (sc->geoms[i].type < PT_GEOM_SPHERE || sc->geoms[i].type > PT_GEOM_TRIANGLE)
      return fail("pt_init: geom %d has unknown type %d", i, sc->geoms[i].type);
  }
  PtOptions opt{};
  if (opt_in) opt = *opt_in;
  PtContext* c = new PtContext();
  if (setup(*c, sc, opt)) {  // nothing half-initialised survives a failure
    destroy(c);
    return -1;
  }
  *out = c;
  return 0;
}

int pt_ctx_destroy(PtContext* c) {
  if (c == g_default) g_default = nullptr;
  destroy(c);
  return 0;
}

int pt_ctx_render(PtContext* c, int iter_first, int iter_count) {
  if (need(c, "pt_render")) return -1;
  if (iter_count <= 0) return 0;
  Ctx& g = *c;
  HIP_OK(hipSetDevice(g.device));
  EventPair ev{};
  if (get_events(g, &ev)) return -1;
  HIP_OK(hipEventRecord(ev.a, g.stream));
  const int end = iter_first + iter_count;
  if (g.failed) return fail("pt_render: an earlier batch of this context failed (%s); free it and create a new one", g_err.c_str());
  for (int it = iter_first; it < end; it += g.K)
    if (run_batch(g, it, std::min(g.K, end - it))) {
      g.failed = true;
      return -1;
    }
  HIP_OK(hipEventRecord(ev.b, g.stream));
  g.pending_render.push_back(ev);
  return 0;
}

int pt_ctx_sync(PtContext* c) {
  if (need(c, "pt_sync")) return -1;
  HIP_OK(hipSetDevice(c->device));
  HIP_OK(hipStreamSynchronize(c->stream));
  return resolve_events(*c);
}

int pt_ctx_readback(PtContext* c, float* out) {
  if (need(c, "pt_readback")) return -1;
  if (!out) return fail("pt_readback: null buffer");
  HIP_OK(hipSetDevice(c->device));
  HIP_OK(hipMemcpyAsync(out, c->d_image, 3 * (size_t)c->N * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  return pt_ctx_sync(c);
}

int pt_ctx_readback_device(PtContext* c, void* out) {
  if (need(c, "pt_readback_device")) return -1;
  if (!out) return fail("pt_readback_device: null buffer");
  HIP_OK(hipSetDevice(c->device));
  HIP_OK(hipMemcpyAsync(out, c->d_image, 3 * (size_t)c->N * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
  return pt_ctx_sync(c);
}

const float* pt_ctx_device_image(PtContext* c) { return c ? c->d_image : nullptr; }
void* pt_ctx_stream(PtContext* c) { return c ? (void*)c->stream : nullptr; }
int pt_ctx_pixel_count(const PtContext* c) { return c ? c->N : 0; }
int pt_ctx_device(const PtContext* c) { return c ? c->device : -1; }

// saveImage's conversion on the device; leaves the bytes in a device buffer owned by the context.
int pt_ctx_save_u8_device(PtContext* c, float samples, const uint8_t** rgb8_dev) {
  if (need(c, "pt_save_u8")) return -1;
  Ctx& g = *c;
  const int W = g.cam.resolution[0];
  if (!(samples > 0.0f)) return fail("pt_save_u8: samples must be positive");
  if (g.pixel_begin % W || g.N % W || (g.stripe && g.stripe != W))
    return fail("pt_save_u8: the tile must consist of whole image rows (begin %d, count %d, stripe %d, width %d)", g.pixel_begin, g.N, g.stripe, W);
  HIP_OK(hipSetDevice(g.device));
  if (!g.d_rgb8 && dalloc(g, &g.d_rgb8, 3 * (size_t)g.N)) return -1;
  g.k->save_u8(g.stream, g.N, W, samples, g.d_image, g.d_rgb8);
  HIP_OK(hipGetLastError());
  if (rgb8_dev) *rgb8_dev = g.d_rgb8;
  return 0;
}
int pt_ctx_save_u8(PtContext* c, float samples, uint8_t* rgb8_host) {
  if (!rgb8_host) return fail("pt_save_u8: null buffer");
  const uint8_t* d = nullptr;
  if (pt_ctx_save_u8_device(c, samples, &d)) return -1;
  HIP_OK(hipMemcpyAsync(rgb8_host, d, 3 * (size_t)c->N, hipMemcpyDeviceToHost, c->stream));
  return pt_ctx_sync(c);
}

int pt_ctx_preview_rgba8_device(PtContext* c, int iterations, void* rgba_dev) {
  if (need(c, "pt_preview_rgba8_device")) return -1;
  if (!rgba_dev || iterations <= 0) return fail("pt_preview_rgba8_device: bad argument");
  HIP_OK(hipSetDevice(c->device));
  c->k->preview(c->stream, c->N, iterations, c->d_image, reinterpret_cast<uchar4*>(rgba_dev));
  HIP_OK(hipStreamSynchronize(c->stream));
  return 0;
}

int pt_ctx_preview_rgba8(PtContext* c, int iterations, uint8_t* rgba_host) {
  if (need(c, "pt_preview_rgba8")) return -1;
  if (!rgba_host || iterations <= 0) return fail("pt_preview_rgba8: bad argument");
  HIP_OK(hipSetDevice(c->device));
  uchar4* d = nullptr;
  HIP_OK(hipMalloc((void**)&d, (size_t)c->N * 4));
  c->k->preview(c->stream, c->N, iterations, c->d_image, d);
  hipError_t e = hipMemcpyAsync(rgba_host, d, (size_t)c->N * 4, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(d);
  if (e != hipSuccess) return fail("pt_preview_rgba8: %s", hipGetErrorString(e));
  return 0;
}

int pt_ctx_get_stats(PtContext* c, PtStats* out) {
  if (need(c, "pt_get_stats")) return -1;
  if (!out) return fail("pt_get_stats: null");
  if (pt_ctx_sync(c)) return -1;
  Ctx& g = *c;
  std::memset(out, 0, sizeof(*out));
  unsigned long long st[PT_MAX_DEPTH];
  HIP_OK(hipMemcpy(st, g.d_stats, sizeof(st), hipMemcpyDeviceToHost));
  for (int d = 0; d < PT_MAX_DEPTH; ++d) out->live_rays[d] = (int64_t)st[d];
  out->samples = g.samples;
  out->intersect_launches = g.isect_launches;
  out->intersect_ms = g.isect_ms;
  out->render_ms = g.render_ms;
  out->num_cus = g.num_cus;
  out->grid_blocks = g.grid_paths > 0 ? g.grid_paths : g.grid_isect;
  out->num_queues = g.qs.Q;
  out->iters_per_batch = g.K;
  out->device_bytes = g.device_bytes;
  out->primary_fused = g.fuse_primary ? 1 : 0;
  out->bounces_fused = g.fuse_bounces ? 1 : 0;
  out->arith = g.arith;
  out->grid_cells = (tables(g).use_grid && g.fuse_bounces) ? g.grid_res[0] * g.grid_res[1] * g.grid_res[2] : 0;
  out->tight_leaves = g.tight_leaves;
  out->paths_waves = 0;
  if (g.qs.deal && g.grid_paths > 0) {  // the table the NEXT batch's k_paths launch will read (ptd::Queues::deal)
    std::vector<int32_t> first((size_t)g.qs.Q + 1);
    HIP_OK(hipMemcpy(first.data(), g.qs.deal, first.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (first[g.qs.Q] == g.qs.paths_W) {
      int lo = first[1] - first[0], hi = lo;
      for (int q = 1; q < g.qs.Q; ++q) lo = std::min(lo, first[q + 1] - first[q]), hi = std::max(hi, first[q + 1] - first[q]);
      out->paths_waves = std::min(lo, 0xffff) << 16 | std::min(hi, 0xffff);
    }
  }
  return 0;
}

int pt_ctx_reset_stats(PtContext* c) {
  if (need(c, "pt_reset_stats")) return -1;
  if (pt_ctx_sync(c)) return -1;
  Ctx& g = *c;
  // on the render stream (created non-blocking: the null stream would not be ordered against it)
  HIP_OK(hipMemsetAsync(g.d_stats, 0, PT_MAX_DEPTH * sizeof(unsigned long long), g.stream));
  HIP_OK(hipStreamSynchronize(g.stream));
  g.samples = 0;
  g.isect_ms = g.render_ms = 0;
  g.isect_launches = 0;
  return 0;
}

int pt_ctx_clear(PtContext* c) {
  if (need(c, "pt_clear")) return -1;
  if (pt_ctx_sync(c)) return -1;
  HIP_OK(hipMemsetAsync(c->d_image, 0, 3 * (size_t)c->N * sizeof(float), c->stream));
  return pt_ctx_reset_stats(c);
}

// ---- the reference's single-instance API (pathtrace.h) on a default context -------------
int pt_free(void) {  // pathtraceFree() before init / twice is legal (main.cpp:134)
  PtContext* c = g_default;
  g_default = nullptr;
  destroy(c);
  return 0;
}
int pt_init(const PtSceneDesc* sc, const PtOptions* opt) {
  pt_free();
  return pt_ctx_create(sc, opt, &g_default);
}
int pt_render(int iter_first, int iter_count) { return pt_ctx_render(g_default, iter_first, iter_count); }
int pt_sync(void) { return pt_ctx_sync(g_default); }
int pt_readback(float* out) { return pt_ctx_readback(g_default, out); }
int pt_readback_device(void* out) { return pt_ctx_readback_device(g_default, out); }
int pt_save_u8(float samples, uint8_t* rgb8_host) { return pt_ctx_save_u8(g_default, samples, rgb8_host); }
int pt_preview_rgba8(int iterations, uint8_t* rgba_host) { return pt_ctx_preview_rgba8(g_default, iterations, rgba_host); }
int pt_preview_rgba8_device(int iterations, void* rgba_dev) { return pt_ctx_preview_rgba8_device(g_default, iterations, rgba_dev); }
int pt_get_stats(PtStats* out) { return pt_ctx_get_stats(g_default, out); }
int pt_reset_stats(void) { return pt_ctx_reset_stats(g_default); }
int pt_clear(void) { return pt_ctx_clear(g_default); }

// ---- stage entry points (tests; default context) ------------------------------------------------
// The C ABI of the stages takes plain SoA float arrays ([3][n]); the kernels stream three planes of 16-byte path records
// (ptd::PathBuf), so the stage wrappers pack / unpack on the host.
namespace {
struct StagePaths {
  ptd::PathBuf pb{};
  std::vector<ptd::Word4> host;
  size_t cap = 0;
  bool alloc(Scratch& sc, size_t cap_) {
    cap = cap_;
    pb.stride = (int64_t)cap;
    pb.r = sc.get<ptd::Word4>(3 * cap);
    host.assign(3 * cap, ptd::Word4{0.f, 0.f, 0.f, 0.f});
    return pb.r != nullptr;
  }
  void pack(int n, const float* o, const float* d, const float* c) {  // arrays are [3][n]; null = zeros
    for (int i = 0; i < n; ++i) {
      auto at = [&](const float* a, int k) { return a ? a[(size_t)k * n + i] : 0.0f; };
      host[i] = ptd::Word4{at(o, 0), at(o, 1), at(o, 2), at(d, 0)};
      host[cap + i] = ptd::Word4{at(d, 1), at(d, 2), at(c, 0), at(c, 1)};
      reinterpret_cast<float*>(&host[2 * cap])[(size_t)i * (ptd::kPathPlane2Bytes / 4)] = at(c, 2);
    }
  }
  int upload() { return hipMemcpy(pb.r, host.data(), host.size() * sizeof(ptd::Word4), hipMemcpyHostToDevice) == hipSuccess ? 0 : -1; }
  int download() { return hipMemcpy(host.data(), pb.r, host.size() * sizeof(ptd::Word4), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1; }
  void unpack(int n, float* o, float* d, float* c) const {
    for (int i = 0; i < n; ++i) {
      const ptd::Word4 &w0 = host[i], &w1 = host[cap + i];
      const float cz = reinterpret_cast<const float*>(&host[2 * cap])[(size_t)i * (ptd::kPathPlane2Bytes / 4)];
      if (o) o[i] = w0.x, o[(size_t)n + i] = w0.y, o[2 * (size_t)n + i] = w0.z;
      if (d) d[i] = w0.w, d[(size_t)n + i] = w1.x, d[2 * (size_t)n + i] = w1.y;
      if (c) c[i] = w1.z, c[(size_t)n + i] = w1.w, c[2 * (size_t)n + i] = cz;
    }
  }
};
}  // namespace

int pt_stage_generate(int pix_begin, int n, float* origin, float* dir) {
  if (need(g_default, "pt_stage_generate")) return -1;
  Ctx& g = *g_default;
  if (n <= 0) return 0;
  HIP_OK(hipSetDevice(g.device));
  Scratch sc;
  ptd::Queues qs = single_queue(g, n);
  StagePaths sp;
  int32_t* cnt = sc.get<int32_t>(16);
  if (!sp.alloc(sc, (size_t)qs.cap) || !cnt) return fail("pt_stage_generate: out of device memory");
  const ptd::PathBuf pb = sp.pb;
  ptk::BatchInfo b{};
  b.iter_first = 1, b.K = 1, b.N = n, b.pixel_begin = pix_begin, b.trace_depth = g.depth;
  b.slot_shift = 30;
  b.aa_jitter = g.aa_jitter ? 1 : 0;
  g.k->generate(g.stream, g.grid, g.dcam, b, qs, pb, cnt);
  HIP_OK(hipStreamSynchronize(g.stream));
  if (sp.download()) return fail("pt_stage_generate: download failed");
  sp.unpack(n, origin, dir, nullptr);
  return 0;
}

int pt_stage_intersect(int n, const float* origin, const float* dir, float* t, float* normal, int32_t* material,
                       float* point) {
  if (need(g_default, "pt_stage_intersect")) return -1;
  Ctx& g = *g_default;
  if (n <= 0) return 0;
  if (g.tight_leaves > 0) {
    // the tightened sphere boxes of a large scene are sized for ray origins inside the scene bounds or at the camera
    // (sphere_tight_box): rays from elsewhere could lose grazing hits, so they are refused rather than traced differently
    double olo[3], ohi[3];
    origin_region(g.root_min, g.root_max, g.cam.position, olo, ohi);
    for (int i = 0; i < n; ++i)
      for (int a = 0; a < 3; ++a) {
        const float v = origin[(size_t)a * n + i];
        if (!(v >= olo[a] && v <= ohi[a]))
          return fail("pt_stage_intersect: ray %d starts outside the scene bounds (this scene's sphere leaves are tightened for origins inside them; "
                      "PtOptions.debug_flags 2048 keeps the reference's boxes)", i);
      }
  }
  HIP_OK(hipSetDevice(g.device));
  Scratch sc;
  ptd::Queues qs = single_queue(g, n);
  const size_t cap = qs.cap;
  StagePaths sp;
  const bool have_paths = sp.alloc(sc, cap);
  const ptd::PathBuf pb = sp.pb;
  ptd::HitBuf hb{};
  hb.stride = cap;
  hb.t = sc.get<float>(cap);
  hb.n = sc.get<float>(3 * cap);
  hb.mat = sc.get<int32_t>(cap);
  hb.p = sc.get<float>(3 * cap);
  int32_t* cnt = sc.get<int32_t>(16);
  if (!have_paths || !hb.t || !hb.n || !hb.mat || !hb.p || !cnt) return fail("pt_stage_intersect: out of device memory");
  sp.pack(n, origin, dir, nullptr);
  if (sp.upload()) return fail("pt_stage_intersect: upload failed");
  HIP_OK(hipMemcpy(cnt, &n, 4, hipMemcpyHostToDevice));
  g.k->intersect(g.stream, g.grid, tables(g), qs, cnt, pb, hb, g.legacy, false);
  HIP_OK(hipStreamSynchronize(g.stream));
  HIP_OK(hipMemcpy(t, hb.t, (size_t)n * 4, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(material, hb.mat, (size_t)n * 4, hipMemcpyDeviceToHost));
  for (int c = 0; c < 3; ++c) {
    HIP_OK(hipMemcpy(normal + (size_t)c * n, hb.n + c * cap, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(point + (size_t)c * n, hb.p + c * cap, (size_t)n * 4, hipMemcpyDeviceToHost));
  }
  return 0;
}

int pt_stage_shade(int n, int depth, const int32_t* iter, const int32_t* pixel, const float* t, const float* normal,
                   const int32_t* material, const float* point, float* origin, float* dir, float* color,
                   int32_t* alive) {
  if (need(g_default, "pt_stage_shade")) return -1;
  Ctx& g = *g_default;
  if (n <= 0) return 0;
  if (depth < 0 || depth >= g.depth) return fail("pt_stage_shade: depth %d outside [0,%d)", depth, g.depth);
  for (int i = 0; i < n; ++i)
    if (t[i] >= 0.0f && (material[i] < 0 || material[i] >= (int)g.mats.size()))
      return fail("pt_stage_shade: material id %d out of range at %d", material[i], i);
  HIP_OK(hipSetDevice(g.device));
  Scratch sc;
  const size_t cap = n;
  StagePaths sp;
  const bool have_paths = sp.alloc(sc, cap);
  const ptd::PathBuf pb = sp.pb;
  ptd::HitBuf hb{};
  hb.stride = cap;
  hb.t = sc.get<float>(cap);
  hb.n = sc.get<float>(3 * cap);
  hb.mat = sc.get<int32_t>(cap);
  hb.p = sc.get<float>(3 * cap);
  int32_t* d_iter = sc.get<int32_t>(cap);
  int32_t* d_pix = sc.get<int32_t>(cap);
  int32_t* d_alive = sc.get<int32_t>(cap);
  if (!have_paths || !hb.t || !hb.n || !hb.mat || !hb.p || !d_iter || !d_pix || !d_alive)
    return fail("pt_stage_shade: out of device memory");
  const size_t b1 = (size_t)n * 4, b3 = 3 * b1;
  sp.pack(n, origin, dir, color);
  if (sp.upload()) return fail("pt_stage_shade: upload failed");
  HIP_OK(hipMemcpy(hb.t, t, b1, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(hb.n, normal, b3, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(hb.mat, material, b1, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(hb.p, point, b3, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(d_iter, iter, b1, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(d_pix, pixel, b1, hipMemcpyHostToDevice));
  g.k->shade_stage(g.stream, tables(g), g.depth, depth, n, d_iter, d_pix, hb, pb, d_alive);
  HIP_OK(hipStreamSynchronize(g.stream));
  if (sp.download()) return fail("pt_stage_shade: download failed");
  sp.unpack(n, origin, dir, color);
  HIP_OK(hipMemcpy(alive, d_alive, b1, hipMemcpyDeviceToHost));
  return 0;
}

// k_save_u8 on a caller-supplied SUM image of whole rows (tests: pins the device conversion to the reference writer's bytes)
int pt_stage_save_u8(int w, int h, float samples, const float* rgb_sum, uint8_t* rgb8) {
  if (need(g_default, "pt_stage_save_u8")) return -1;
  Ctx& g = *g_default;
  if (w <= 0 || h <= 0 || h >= 32768 || !rgb_sum || !rgb8 || !(samples > 0.0f)) return fail("pt_stage_save_u8: bad argument");
  HIP_OK(hipSetDevice(g.device));
  Scratch sc;
  const size_t n = (size_t)w * h;
  float* d_img = sc.get<float>(3 * n);
  uint8_t* d_u8 = sc.get<uint8_t>(3 * n);
  if (!d_img || !d_u8) return fail("pt_stage_save_u8: out of device memory");
  HIP_OK(hipMemcpy(d_img, rgb_sum, 3 * n * sizeof(float), hipMemcpyHostToDevice));
  g.k->save_u8(g.stream, (int)n, w, samples, d_img, d_u8);
  HIP_OK(hipStreamSynchronize(g.stream));
  HIP_OK(hipMemcpy(rgb8, d_u8, 3 * n, hipMemcpyDeviceToHost));
  return 0;
}

}  // extern "C"
