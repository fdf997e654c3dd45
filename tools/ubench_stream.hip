// Micro-benchmark: persistent-grid streaming of path records with different layouts.
//  mode 0: 6 dword plane loads + 8 dword plane stores per lane (SoA of scalars, current layout)
//  mode 1: float3 + float3 loads, float4 + float4 stores (SoA of small vectors)
//  mode 2: float4 + float2 loads, float4 + float4 stores
//  modes 0-2 move 56 B per element.
//  mode 3 (round 3): the fused bounce kernel's own record layout and traffic — three planes of 16 + 16 + 8 B read per path,
//          and per path either the same three planes written (a survivor, 71 % of the lanes — cornell's ray-weighted average over
//          depths 1-7 — contiguous per wave after a ballot / prefix compaction, like the kernel's queue append) or one 16-B record
//          (a retired sample): 40 B read + 0.71 * 40 + 0.29 * 16 = 33 B written per path; nothing else is done with the data.  What a kernel with
//          k_bounce's bytes and access shape can reach on this machine when it does no work at all.
// Reports TB/s for N elements.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <string>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
struct __attribute__((packed, aligned(4))) f3 { float x, y, z; };
template <int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ in, float* __restrict__ out, long n, long S) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    if (MODE == 0) {
      float a = in[i], b = in[S + i], c = in[2 * S + i], d = in[3 * S + i], e = in[4 * S + i], f = in[5 * S + i];
      float s = a + b + c + d + e + f;
      for (int p = 0; p < 8; ++p) out[p * S + i] = s + p;
    } else if (MODE == 1) {
      const f3* O = reinterpret_cast<const f3*>(in);
      const f3* D = reinterpret_cast<const f3*>(in + 3 * S);
      f3 o = O[i], d = D[i];
      float s = o.x + o.y + o.z + d.x + d.y + d.z;
      float4* H0 = reinterpret_cast<float4*>(out);
      float4* H1 = reinterpret_cast<float4*>(out + 4 * S);
      H0[i] = make_float4(s, s + 1, s + 2, s + 3);
      H1[i] = make_float4(s + 4, s + 5, s + 6, s + 7);
    } else {
      const float4* R0 = reinterpret_cast<const float4*>(in);
      const float2* R1 = reinterpret_cast<const float2*>(in + 4 * S);
      float4 a = R0[i]; float2 b = R1[i];
      float s = a.x + a.y + a.z + a.w + b.x + b.y;
      float4* H0 = reinterpret_cast<float4*>(out);
      float4* H1 = reinterpret_cast<float4*>(out + 4 * S);
      H0[i] = make_float4(s, s + 1, s + 2, s + 3);
      H1[i] = make_float4(s + 4, s + 5, s + 6, s + 7);
    }
  }
}
struct W4 { float x, y, z, w; };
// V 0: as the kernel does it — Q = 256 queues own n / Q consecutive paths each, the waves w, w + Q, ... of the grid serve queue w % Q
//      (all on one XCD: blockIdx steps by 64), take its 64-path groups round-robin, append the survivors to the queue's output
//      region at a base reserved with one returning atomic per wave and group, and the retired records to a segment of their own;
//   1: the same with the reservation taken from a wave-private counter (no atomic): output runs of one wave stay contiguous,
//      runs of different waves leave gaps;  2: no retired records;  3: every path survives (plain copy in this work distribution)
template <int V>
__global__ __launch_bounds__(256) void k3(const W4* __restrict__ in, const float2* __restrict__ in2, W4* __restrict__ out, float2* __restrict__ out2,
                                          W4* __restrict__ ret, int* __restrict__ cnt, long n, int Q) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), W = gridDim.x * 4;
  const int q = wave % Q, r = wave / Q, wq = W / Q;
  const long cap = n / Q, qbase = (long)q * cap;  // n is a multiple of 64 * Q
  long mine = 0, dead_at = 0;
  for (long j = r; j * 64 < cap; j += wq) {
    const long at = qbase + j * 64 + lane;
    const W4 a = in[at], b = in[n + at];
    const float2 c = in2[at];
    const bool live = V == 3 || ((__float_as_uint(a.x) ^ (uint32_t)at * 2654435761u) % 100u) < 71u;
    const unsigned long long m = __ballot(live);
    const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
    long base;
    if (V == 0) {
      int bq = 0;
      if (lane == 0) bq = atomicAdd(&cnt[q * 16], __popcll(m));
      base = qbase + __builtin_amdgcn_readfirstlane(bq);
    } else {
      base = qbase + (long)r * (cap / wq) + mine;  // the wave's own slice of the queue's region
      mine += __popcll(m);
    }
    if (live) {
      const long o = base + rank;
      out[o] = W4{a.x + 1, a.y, a.z, b.x}, out[n + o] = W4{b.y, b.z, b.w, c.x}, out2[o] = make_float2(c.y, a.w);
    } else if (V < 2) {
      ret[qbase + (long)r * (cap / wq) + dead_at + (lane - rank)] = W4{a.x, b.y, c.x, a.w};
    }
    dead_at += 64 - __popcll(m);
  }
}
template <int V> int run3(long n, int bpc, int Q = 256, bool json = false) {
  W4 *in, *out, *ret; float2 *in2, *out2; int* cnt;
  CK(hipMalloc(&in, n * 32 + 64)); CK(hipMalloc(&in2, n * 8 + 64)); CK(hipMalloc(&out, n * 32 + 64)); CK(hipMalloc(&out2, n * 8 + 64));
  CK(hipMalloc(&ret, n * 16 + 1024)); CK(hipMalloc(&cnt, 65536));
  n = n / (64 * 1024) * (64 * 1024);
  CK(hipMemset(in, 1, n * 32)); CK(hipMemset(in2, 1, n * 8)); CK(hipMemset(cnt, 0, 65536));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grid = 256 * bpc;
  for (int w = 0; w < 2; ++w) {
    CK(hipMemsetAsync(cnt, 0, 65536));
    hipLaunchKernelGGL(k3<V>, dim3(grid), dim3(256), 0, 0, in, in2, out, out2, ret, cnt, n, Q);
  }
  const int reps = 10;
  float ms = 0;
  for (int r = 0; r < reps; ++r) {
    float one;
    CK(hipMemsetAsync(cnt, 0, 65536));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k3<V>, dim3(grid), dim3(256), 0, 0, in, in2, out, out2, ret, cnt, n, Q);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&one, e0, e1));
    ms += one;
  }
  const double bytes = V == 3 ? 80.0 : 40.0 + 0.71 * 40.0 + (V == 2 ? 0.0 : 0.29 * 16.0);
  const char* names[4] = {"k_bounce traffic, no work", "  same, wave-private output slices", "  same, no retired records", "  every path survives (copy)"};
  if (json)
    printf("{\"paths\": %ld, \"us_per_launch\": %.2f, \"bytes_per_path\": %.1f, \"tb_per_s\": %.3f, \"queues\": %d, \"blocks_per_cu\": %d}\n", n, ms * 1e3 / reps, bytes,
           bytes * n * reps / (ms * 1e-3) / 1e12, Q, bpc);
  else
    printf("%-34s n=%.1fM blocks/CU=%d Q=%d: %.1f us/launch, %.2f TB/s (%.1f B/path)\n", names[V], n / 1048576.0, bpc, Q, ms * 1e3 / reps,
           bytes * n * reps / (ms * 1e-3) / 1e12, bytes);
  CK(hipFree(in)); CK(hipFree(in2)); CK(hipFree(out)); CK(hipFree(out2)); CK(hipFree(ret)); CK(hipFree(cnt));
  return 0;
}
template <int MODE> int run(const char* name, long n, int bpc) {
  float *in, *out;
  CK(hipMalloc(&in, n * 6 * 4 + 64)); CK(hipMalloc(&out, n * 8 * 4 + 64));
  CK(hipMemset(in, 0, n * 6 * 4)); 
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int grid = 256 * bpc;
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, in, out, n, n);
  CK(hipEventRecord(e0));
  const int reps = 10;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, in, out, n, n);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-34s n=%ldM blocks/CU=%d: %.1f us/launch, %.2f TB/s\n", name, n >> 20, bpc, ms * 1e3 / reps, 56.0 * n * reps / (ms * 1e-3) / 1e12);
  CK(hipFree(in)); CK(hipFree(out));
  return 0;
}
int main(int argc, char** argv) {
  // --floor N: only the no-work kernel with k_bounce's traffic on N paths (256 queues, 4 blocks per CU), one JSON line (bench.py)
  if (argc >= 3 && std::string(argv[1]) == "--floor") return run3<0>(atol(argv[2]), 4, 256, true);
  for (int Q : {16, 64, 256, 1024}) {  // number of queues = concurrent input / output streams
    if (run3<0>(11630000L, 4, Q)) return 1;
    if (run3<3>(11630000L, 4, Q)) return 1;
  }
  for (long n : {24100000L, 11630000L}) for (int bpc : {4, 8}) {
    if (run3<0>(n, bpc)) return 1;
    if (run3<1>(n, bpc)) return 1;
    if (run3<2>(n, bpc)) return 1;
    if (run3<3>(n, bpc)) return 1;
  }
  for (long n : {12L << 20, 3L << 20}) for (int bpc : {5, 8}) {
    if (run<0>("SoA dword (6 ld + 8 st)", n, bpc)) return 1;
    if (run<1>("float3,float3 -> float4,float4", n, bpc)) return 1;
    if (run<2>("float4,float2 -> float4,float4", n, bpc)) return 1;
  }
  return 0;
}
