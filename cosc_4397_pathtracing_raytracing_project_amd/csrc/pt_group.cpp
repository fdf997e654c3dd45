// pt_group.cpp — one process, several MI355X: BASELINE config 4 (framebuffer tiled across the GPUs of a node, a
// single RCCL gather over xGMI at image write-out; SURVEY.md §8e, §5 "Distributed communication backend").
//
// The reference is single-device (src/preview.cpp:112 cudaGLSetGLDevice(0)); its write-out point is saveImage()
// (src/main.cpp:86-107).  Here every device owns the rows i, i+n, i+2n, ... of the frame (work per row varies
// smoothly down the cornell frame, interleaving gives every device the same mix) and runs ALL iterations on them
// on its own stream; samples are keyed by the global pixel index (makeSeededRandomEngine(iter, idx, depth),
// src/pathtrace.cu:203-207,368), so nothing is ever summed across devices and the assembled image is bit-identical
// to the single-GPU image.  The only communication is, once per write-out, one grouped ncclSend/ncclRecv of the
// tiles into devices[0] on a ncclCommInitAll communicator — xGMI is point-to-point, the tiles go straight to the
// root over their own links, there is no ring and no reduction.  Placement of the interleaved rows (a strided 2-D
// copy on the root device) and one D2H copy follow.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <vector>

#include "../../include/pt_amd.h"
#include "pt_internal.h"

#define HIP_OK(expr)                                                                                      \
  do {                                                                                                    \
    hipError_t e_ = (expr);                                                                               \
    if (e_ != hipSuccess) return pt_fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)
#define NCCL_OK(expr)                                                                                      \
  do {                                                                                                     \
    ncclResult_t r_ = (expr);                                                                              \
    if (r_ != ncclSuccess) return pt_fail("%s failed: %s (%s:%d)", #expr, ncclGetErrorString(r_), __FILE__, __LINE__); \
  } while (0)

struct PtGroup {
  int W = 0, H = 0, n = 0;
  std::vector<int> devices;
  std::vector<PtContext*> ctx;
  std::vector<ncclComm_t> comms;
  std::vector<int> rows;         // image rows owned by device i
  std::vector<size_t> recv_off;  // pixel offset of device i's tile in the root's receive buffer (i >= 1)
  size_t recv_pixels = 0;
  float* d_recv = nullptr;    // root: tiles of devices 1..n-1, float RGB
  float* d_full = nullptr;    // root: assembled frame
  uint8_t* d_recv8 = nullptr; // same for the converted bytes (3 B/pixel)
  uint8_t* d_full8 = nullptr;
  std::vector<uint8_t*> d_prev;  // per device: RGBA8 preview of its tile (sendImageToPBO)
};

namespace {

void release(PtGroup* g) {
  if (!g) return;
  for (size_t i = 0; i < g->comms.size(); ++i)
    if (g->comms[i]) {
      (void)hipSetDevice(g->devices[i]);
      (void)ncclCommDestroy(g->comms[i]);
    }
  if (!g->devices.empty()) (void)hipSetDevice(g->devices[0]);
  for (void* p : {(void*)g->d_recv, (void*)g->d_full, (void*)g->d_recv8, (void*)g->d_full8})
    if (p) (void)hipFree(p);
  for (size_t i = 0; i < g->d_prev.size(); ++i)
    if (g->d_prev[i]) {
      (void)hipSetDevice(g->devices[i]);
      (void)hipFree(g->d_prev[i]);
    }
  for (PtContext* c : g->ctx) (void)pt_ctx_destroy(c);
  delete g;
}

// One grouped send/recv: device i >= 1 sends `elems_per_pixel * pixels_i` elements of `src(i)` on its own stream,
// the root receives them back to back on its stream.
template <typename T, typename SrcFn>
int exchange(PtGroup* g, ncclDataType_t type, int elems_per_pixel, T* root_recv, SrcFn src) {
  if (g->n == 1) return 0;
  NCCL_OK(ncclGroupStart());
  int rc = 0;
  for (int i = 1; i < g->n && !rc; ++i) {  // an error inside the group must still close it
    const size_t count = (size_t)elems_per_pixel * pt_ctx_pixel_count(g->ctx[i]);
    ncclResult_t r = ncclSuccess;
    if (hipSetDevice(g->devices[i]) != hipSuccess) rc = pt_fail("hipSetDevice(%d) failed", g->devices[i]);
    else if ((r = ncclSend(src(i), count, type, 0, g->comms[i], (hipStream_t)pt_ctx_stream(g->ctx[i]))) != ncclSuccess)
      rc = pt_fail("ncclSend from device %d failed: %s", g->devices[i], ncclGetErrorString(r));
    else if (hipSetDevice(g->devices[0]) != hipSuccess) rc = pt_fail("hipSetDevice(%d) failed", g->devices[0]);
    else if ((r = ncclRecv(root_recv + (size_t)elems_per_pixel * g->recv_off[i], count, type, i, g->comms[0],
                           (hipStream_t)pt_ctx_stream(g->ctx[0]))) != ncclSuccess)
      rc = pt_fail("ncclRecv from device %d failed: %s", g->devices[i], ncclGetErrorString(r));
  }
  const ncclResult_t e = ncclGroupEnd();
  if (!rc && e != ncclSuccess) rc = pt_fail("ncclGroupEnd failed: %s", ncclGetErrorString(e));
  return rc;
}

// Rows of device i (tile order) -> rows i, i+n, ... of the frame, on the root's stream.
template <typename T>
int place_rows(PtGroup* g, int i, const T* tile, T* full, size_t bytes_per_pixel) {
  const size_t row = (size_t)g->W * bytes_per_pixel;
  HIP_OK(hipMemcpy2DAsync(reinterpret_cast<char*>(full) + (size_t)i * row, (size_t)g->n * row, tile, row, row, (size_t)g->rows[i],
                          hipMemcpyDeviceToDevice, (hipStream_t)pt_ctx_stream(g->ctx[0])));
  return 0;
}

}  // namespace

extern "C" {

int pt_group_create(const PtSceneDesc* scene, const PtOptions* base, const int* devices, int num_devices, PtGroup** out) {
  if (!out) return pt_fail("pt_group_create: null output");
  *out = nullptr;
  if (!scene || !devices || num_devices <= 0) return pt_fail("pt_group_create: bad argument");
  const int W = scene->camera.resolution[0], H = scene->camera.resolution[1];
  if (num_devices > H) return pt_fail("pt_group_create: more devices (%d) than image rows (%d)", num_devices, H);
  for (int i = 0; i < num_devices; ++i)
    for (int j = 0; j < i; ++j)
      if (devices[i] == devices[j]) return pt_fail("pt_group_create: device %d listed twice", devices[i]);
  PtGroup* g = new PtGroup();
  g->W = W, g->H = H, g->n = num_devices;
  g->devices.assign(devices, devices + num_devices);
  g->rows.resize(num_devices);
  g->recv_off.assign(num_devices, 0);
  for (int i = 0; i < num_devices; ++i) {
    g->rows[i] = (H - i + num_devices - 1) / num_devices;
    if (i >= 1) {
      g->recv_off[i] = g->recv_pixels;
      g->recv_pixels += (size_t)g->rows[i] * W;
    }
    PtOptions opt{};
    if (base) opt = *base;
    opt.device = devices[i];
    opt.pixel_begin = i * W;
    opt.pixel_count = g->rows[i] * W;
    opt.stripe_pixels = num_devices > 1 ? W : 0;
    opt.stripe_stride = num_devices > 1 ? num_devices * W : 0;
    PtContext* c = nullptr;
    if (pt_ctx_create(scene, &opt, &c)) {
      release(g);
      return -1;
    }
    g->ctx.push_back(c);
  }
  g->comms.assign(num_devices, nullptr);
  ncclResult_t r = ncclCommInitAll(g->comms.data(), num_devices, g->devices.data());
  if (r != ncclSuccess) {
    release(g);
    return pt_fail("ncclCommInitAll(%d devices) failed: %s", num_devices, ncclGetErrorString(r));
  }
  *out = g;
  return 0;
}

int pt_group_destroy(PtGroup* g) {
  release(g);
  return 0;
}
int pt_group_size(const PtGroup* g) { return g ? g->n : 0; }
PtContext* pt_group_context(PtGroup* g, int i) { return (g && i >= 0 && i < g->n) ? g->ctx[i] : nullptr; }

int pt_group_render(PtGroup* g, int iter_first, int iter_count) {
  if (!g) return pt_fail("pt_group_render: null group");
  for (PtContext* c : g->ctx)  // launches are asynchronous: the devices run concurrently
    if (pt_ctx_render(c, iter_first, iter_count)) return -1;
  return 0;
}

int pt_group_sync(PtGroup* g) {
  if (!g) return pt_fail("pt_group_sync: null group");
  for (PtContext* c : g->ctx)
    if (pt_ctx_sync(c)) return -1;
  return 0;
}

int pt_group_gather(PtGroup* g, float* rgb_sum_host) {
  if (!g || !rgb_sum_host) return pt_fail("pt_group_gather: bad argument");
  const size_t frame = (size_t)g->W * g->H;
  HIP_OK(hipSetDevice(g->devices[0]));
  if (g->n == 1) return pt_ctx_readback(g->ctx[0], rgb_sum_host);
  if (!g->d_full) HIP_OK(hipMalloc((void**)&g->d_full, frame * 12));
  if (!g->d_recv) HIP_OK(hipMalloc((void**)&g->d_recv, g->recv_pixels * 12));
  if (exchange(g, ncclFloat, 3, g->d_recv, [&](int i) { return pt_ctx_device_image(g->ctx[i]); })) return -1;
  HIP_OK(hipSetDevice(g->devices[0]));
  for (int i = 0; i < g->n; ++i)
    if (place_rows(g, i, i == 0 ? pt_ctx_device_image(g->ctx[0]) : g->d_recv + 3 * g->recv_off[i], g->d_full, 12)) return -1;
  HIP_OK(hipMemcpyAsync(rgb_sum_host, g->d_full, frame * 12, hipMemcpyDeviceToHost, (hipStream_t)pt_ctx_stream(g->ctx[0])));
  return pt_group_sync(g);
}

int pt_group_gather_u8(PtGroup* g, float samples, uint8_t* rgb8_host) {
  if (!g || !rgb8_host) return pt_fail("pt_group_gather_u8: bad argument");
  const size_t frame = (size_t)g->W * g->H;
  std::vector<const uint8_t*> tiles(g->n, nullptr);
  for (int i = 0; i < g->n; ++i)  // every device converts its own rows (x mirror is inside a row)
    if (pt_ctx_save_u8_device(g->ctx[i], samples, &tiles[i])) return -1;
  HIP_OK(hipSetDevice(g->devices[0]));
  if (g->n == 1) {
    HIP_OK(hipMemcpyAsync(rgb8_host, tiles[0], frame * 3, hipMemcpyDeviceToHost, (hipStream_t)pt_ctx_stream(g->ctx[0])));
    return pt_group_sync(g);
  }
  if (!g->d_full8) HIP_OK(hipMalloc((void**)&g->d_full8, frame * 3));
  if (!g->d_recv8) HIP_OK(hipMalloc((void**)&g->d_recv8, g->recv_pixels * 3));
  if (exchange(g, ncclUint8, 3, g->d_recv8, [&](int i) { return tiles[i]; })) return -1;
  HIP_OK(hipSetDevice(g->devices[0]));
  for (int i = 0; i < g->n; ++i)
    if (place_rows(g, i, i == 0 ? tiles[0] : g->d_recv8 + 3 * g->recv_off[i], g->d_full8, 3)) return -1;
  HIP_OK(hipMemcpyAsync(rgb8_host, g->d_full8, frame * 3, hipMemcpyDeviceToHost, (hipStream_t)pt_ctx_stream(g->ctx[0])));
  return pt_group_sync(g);
}

// Progressive preview (the reference converts and shows the running average after EVERY iteration: sendImageToPBO,
// src/pathtrace.cu:250-268,618): every device converts its own rows (average over `iterations`, gamma 1/2.2, clamp,
// RGBA8), then the same single exchange + row placement as the write-out, 4 B per pixel.  Meant to be called every N
// batches while rendering continues; it synchronises the devices (the image is read at a batch boundary).
int pt_group_preview_rgba8(PtGroup* g, int iterations, uint8_t* rgba_host) {
  if (!g || !rgba_host || iterations <= 0) return pt_fail("pt_group_preview_rgba8: bad argument");
  const size_t frame = (size_t)g->W * g->H;
  if (g->d_prev.empty()) g->d_prev.assign(g->n, nullptr);
  for (int i = 0; i < g->n; ++i) {
    HIP_OK(hipSetDevice(g->devices[i]));
    if (!g->d_prev[i]) HIP_OK(hipMalloc((void**)&g->d_prev[i], (size_t)pt_ctx_pixel_count(g->ctx[i]) * 4));
    if (pt_ctx_preview_rgba8_device(g->ctx[i], iterations, g->d_prev[i])) return -1;
  }
  HIP_OK(hipSetDevice(g->devices[0]));
  if (g->n == 1) {
    HIP_OK(hipMemcpyAsync(rgba_host, g->d_prev[0], frame * 4, hipMemcpyDeviceToHost, (hipStream_t)pt_ctx_stream(g->ctx[0])));
    return pt_group_sync(g);
  }
  uint8_t *full = nullptr, *recv = nullptr;  // previews are occasional: scratch buffers, freed below
  HIP_OK(hipMalloc((void**)&full, frame * 4));
  HIP_OK(hipMalloc((void**)&recv, g->recv_pixels * 4));
  int rc = exchange(g, ncclUint8, 4, recv, [&](int i) { return (const uint8_t*)g->d_prev[i]; });
  if (!rc) {
    (void)hipSetDevice(g->devices[0]);
    for (int i = 0; i < g->n && !rc; ++i) rc = place_rows(g, i, i == 0 ? (const uint8_t*)g->d_prev[0] : recv + 4 * g->recv_off[i], full, 4);
  }
  if (!rc && hipMemcpyAsync(rgba_host, full, frame * 4, hipMemcpyDeviceToHost, (hipStream_t)pt_ctx_stream(g->ctx[0])) != hipSuccess)
    rc = pt_fail("pt_group_preview_rgba8: copy failed");
  if (!rc) rc = pt_group_sync(g);
  (void)hipSetDevice(g->devices[0]);
  (void)hipFree(full);
  (void)hipFree(recv);
  return rc;
}

}  // extern "C"
