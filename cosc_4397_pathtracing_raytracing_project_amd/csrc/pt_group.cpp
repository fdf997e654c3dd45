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
//
// Two transports for that one exchange (PtGroup::transport, pt_group_create_ex):
//   RCCL  the grouped ncclSend / ncclRecv above — the default for distinct devices;
//   COPY  hipMemcpyPeerAsync (hipMemcpyAsync when source and root are the same device) of every tile into the same
//         receive buffer at the same offsets, ordered after the tile's producer by an event the root's stream waits
//         on.  RCCL cannot put two ranks of one communicator on one device, so this is what a group whose device
//         list names a device more than once (e.g. {0, 0, 0}: several contexts, one GPU) uses automatically; it
//         makes the whole multi-context path — per-context streams, receive offsets, strided row placement,
//         preview exchange — testable on a one-GPU box, and is a fallback where RCCL is unwanted.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <string>
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
  int transport = PT_GROUP_TRANSPORT_RCCL;  // resolved (never AUTO) after pt_group_create_ex
  std::vector<hipEvent_t> ready;            // COPY transport: "tile i is complete" (recorded on context i's stream)
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
  uint8_t* d_recv_prev = nullptr;  // root: preview tiles of devices 1..n-1 / assembled preview (kept: previews recur)
  uint8_t* d_full_prev = nullptr;
};

namespace {

void release(PtGroup* g) {
  if (!g) return;
  for (size_t i = 0; i < g->comms.size(); ++i)
    if (g->comms[i]) {
      (void)hipSetDevice(g->devices[i]);
      (void)ncclCommDestroy(g->comms[i]);
    }
  for (size_t i = 0; i < g->ready.size(); ++i)
    if (g->ready[i]) {
      (void)hipSetDevice(g->devices[i]);
      (void)hipEventDestroy(g->ready[i]);
    }
  if (!g->devices.empty()) (void)hipSetDevice(g->devices[0]);
  for (void* p : {(void*)g->d_recv, (void*)g->d_full, (void*)g->d_recv8, (void*)g->d_full8, (void*)g->d_recv_prev,
                  (void*)g->d_full_prev})
    if (p) (void)hipFree(p);
  for (size_t i = 0; i < g->d_prev.size(); ++i)
    if (g->d_prev[i]) {
      (void)hipSetDevice(g->devices[i]);
      (void)hipFree(g->d_prev[i]);
    }
  for (PtContext* c : g->ctx) (void)pt_ctx_destroy(c);
  delete g;
}

// COPY transport: the root's stream waits for tile i's producer, then pulls the tile into the receive buffer.
int exchange_copy(PtGroup* g, size_t bytes_per_pixel, char* root_recv, const void* const* src) {
  hipStream_t root = (hipStream_t)pt_ctx_stream(g->ctx[0]);
  for (int i = 1; i < g->n; ++i) {
    const size_t bytes = bytes_per_pixel * (size_t)pt_ctx_pixel_count(g->ctx[i]);
    HIP_OK(hipSetDevice(g->devices[i]));
    HIP_OK(hipEventRecord(g->ready[i], (hipStream_t)pt_ctx_stream(g->ctx[i])));
    HIP_OK(hipSetDevice(g->devices[0]));
    HIP_OK(hipStreamWaitEvent(root, g->ready[i], 0));
    char* dst = root_recv + bytes_per_pixel * g->recv_off[i];
    if (g->devices[i] == g->devices[0]) HIP_OK(hipMemcpyAsync(dst, src[i], bytes, hipMemcpyDeviceToDevice, root));
    else HIP_OK(hipMemcpyPeerAsync(dst, g->devices[0], src[i], g->devices[i], bytes, root));
  }
  return 0;
}

// The one exchange of a write-out: device i >= 1 hands `elems_per_pixel * pixels_i` elements of `src(i)` to the root,
// which receives them back to back (recv_off) on its stream.  RCCL: one grouped send/recv, each send on its own
// device's stream.
template <typename T, typename SrcFn>
int exchange(PtGroup* g, ncclDataType_t type, int elems_per_pixel, T* root_recv, SrcFn src) {
  if (g->n == 1) return 0;
  if (g->transport == PT_GROUP_TRANSPORT_COPY) {
    std::vector<const void*> srcs(g->n, nullptr);
    for (int i = 1; i < g->n; ++i) srcs[i] = src(i);
    return exchange_copy(g, sizeof(T) * (size_t)elems_per_pixel, reinterpret_cast<char*>(root_recv), srcs.data());
  }
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

// After a failed exchange / placement some sends, receives or copies may already be queued on the devices' streams:
// drain them (ignoring further errors, keeping the first message) before the caller sees the failure.
int fail_after_drain(PtGroup* g) {
  const std::string first = pt_last_error();
  for (PtContext* c : g->ctx) (void)pt_ctx_sync(c);
  return pt_fail("%s", first.c_str());
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

int pt_group_create_ex(const PtSceneDesc* scene, const PtOptions* base, const int* devices, int num_devices, int transport,
                       PtGroup** out) {
  if (!out) return pt_fail("pt_group_create: null output");
  *out = nullptr;
  if (!scene || !devices || num_devices <= 0) return pt_fail("pt_group_create: bad argument");
  if (transport != PT_GROUP_TRANSPORT_AUTO && transport != PT_GROUP_TRANSPORT_RCCL && transport != PT_GROUP_TRANSPORT_COPY)
    return pt_fail("pt_group_create: unknown transport %d", transport);
  const int W = scene->camera.resolution[0], H = scene->camera.resolution[1];
  if (num_devices > H) return pt_fail("pt_group_create: more devices (%d) than image rows (%d)", num_devices, H);
  bool shared = false;  // a device named more than once: several contexts on one GPU
  for (int i = 0; i < num_devices; ++i)
    for (int j = 0; j < i; ++j) shared |= devices[i] == devices[j];
  if (shared && transport == PT_GROUP_TRANSPORT_RCCL)
    return pt_fail("pt_group_create: a device is listed twice; RCCL cannot place two ranks of one communicator on one "
                   "device (use PT_GROUP_TRANSPORT_AUTO or _COPY)");
  PtGroup* g = new PtGroup();
  g->W = W, g->H = H, g->n = num_devices;
  g->transport = transport == PT_GROUP_TRANSPORT_AUTO ? (shared ? PT_GROUP_TRANSPORT_COPY : PT_GROUP_TRANSPORT_RCCL) : transport;
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
  if (g->transport == PT_GROUP_TRANSPORT_COPY) {
    g->ready.assign(num_devices, nullptr);
    for (int i = 1; i < num_devices; ++i)
      if (hipSetDevice(devices[i]) != hipSuccess || hipEventCreateWithFlags(&g->ready[i], hipEventDisableTiming) != hipSuccess) {
        release(g);
        return pt_fail("pt_group_create: cannot create the tile-ready event on device %d", devices[i]);
      }
  } else {
    g->comms.assign(num_devices, nullptr);
    ncclResult_t r = ncclCommInitAll(g->comms.data(), num_devices, g->devices.data());
    if (r != ncclSuccess) {
      release(g);
      return pt_fail("ncclCommInitAll(%d devices) failed: %s", num_devices, ncclGetErrorString(r));
    }
  }
  *out = g;
  return 0;
}

int pt_group_create(const PtSceneDesc* scene, const PtOptions* base, const int* devices, int num_devices, PtGroup** out) {
  return pt_group_create_ex(scene, base, devices, num_devices, PT_GROUP_TRANSPORT_AUTO, out);
}

int pt_group_transport(const PtGroup* g) { return g ? g->transport : -1; }

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
  if (exchange(g, ncclFloat, 3, g->d_recv, [&](int i) { return pt_ctx_device_image(g->ctx[i]); })) return fail_after_drain(g);
  HIP_OK(hipSetDevice(g->devices[0]));
  for (int i = 0; i < g->n; ++i)
    if (place_rows(g, i, i == 0 ? pt_ctx_device_image(g->ctx[0]) : g->d_recv + 3 * g->recv_off[i], g->d_full, 12))
      return fail_after_drain(g);
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
  if (exchange(g, ncclUint8, 3, g->d_recv8, [&](int i) { return tiles[i]; })) return fail_after_drain(g);
  HIP_OK(hipSetDevice(g->devices[0]));
  for (int i = 0; i < g->n; ++i)
    if (place_rows(g, i, i == 0 ? tiles[0] : g->d_recv8 + 3 * g->recv_off[i], g->d_full8, 3)) return fail_after_drain(g);
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
  if (!g->d_full_prev) HIP_OK(hipMalloc((void**)&g->d_full_prev, frame * 4));  // kept in the group: previews recur every
  if (!g->d_recv_prev) HIP_OK(hipMalloc((void**)&g->d_recv_prev, g->recv_pixels * 4));  // few batches; freed by release()
  if (exchange(g, ncclUint8, 4, g->d_recv_prev, [&](int i) { return (const uint8_t*)g->d_prev[i]; })) return fail_after_drain(g);
  HIP_OK(hipSetDevice(g->devices[0]));
  for (int i = 0; i < g->n; ++i)
    if (place_rows(g, i, i == 0 ? (const uint8_t*)g->d_prev[0] : g->d_recv_prev + 4 * g->recv_off[i], g->d_full_prev, 4))
      return fail_after_drain(g);
  HIP_OK(hipMemcpyAsync(rgba_host, g->d_full_prev, frame * 4, hipMemcpyDeviceToHost, (hipStream_t)pt_ctx_stream(g->ctx[0])));
  return pt_group_sync(g);
}

}  // extern "C"
