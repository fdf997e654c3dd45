// Micro-benchmark for the finalGather access pattern: K*3 planes of N floats read with 16-B loads, plane stride Np.
// Question: does a plane stride that is a multiple of 4 KiB (1920*1080*4 B = 2025 * 4096) cost HBM bandwidth?
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ __launch_bounds__(256) void k(const float* __restrict__ fin, float* __restrict__ img, int n4, long Np, int K, int U) {
  const long FS = (long)K * Np;
  for (int q = blockIdx.x * 256 + threadIdx.x; q < n4; q += gridDim.x * 256) {
    const long p = 4L * q;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int k = 0; k < K; k += 4) {
      float4 v[12];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long s = (long)(k + u < K ? k + u : K - 1) * Np + p;
        v[3 * u] = *reinterpret_cast<const float4*>(fin + s);
        v[3 * u + 1] = *reinterpret_cast<const float4*>(fin + FS + s);
        v[3 * u + 2] = *reinterpret_cast<const float4*>(fin + 2 * FS + s);
      }
#pragma unroll
      for (int u = 0; u < 12; ++u) acc.x += v[u].x, acc.y += v[u].y, acc.z += v[u].z, acc.w += v[u].w;
    }
    *reinterpret_cast<float4*>(img + p) = acc;
  }
}

__global__ __launch_bounds__(256) void kg4(const float* __restrict__ final_rgb, float* __restrict__ image, int N, int K) {
  const long FS = (long)K * N;
  const int n4 = N / 4;
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += gridDim.x * blockDim.x) {
    const long p = 4 * (long)q;
    float4* img = reinterpret_cast<float4*>(image + 3 * p);
    const float4 i0 = img[0], i1 = img[1], i2 = img[2];
    float r[4] = {i0.x, i0.w, i1.z, i2.y}, g[4] = {i0.y, i1.x, i1.w, i2.z}, bl[4] = {i0.z, i1.y, i2.x, i2.w};
    int k = 0;
    for (; k + 4 <= K; k += 4) {
      float4 vr[4], vg[4], vb[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long s = (long)(k + u) * N + p;
        vr[u] = *reinterpret_cast<const float4*>(final_rgb + s);
        vg[u] = *reinterpret_cast<const float4*>(final_rgb + FS + s);
        vb[u] = *reinterpret_cast<const float4*>(final_rgb + 2 * FS + s);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        r[0] += vr[u].x, r[1] += vr[u].y, r[2] += vr[u].z, r[3] += vr[u].w;
        g[0] += vg[u].x, g[1] += vg[u].y, g[2] += vg[u].z, g[3] += vg[u].w;
        bl[0] += vb[u].x, bl[1] += vb[u].y, bl[2] += vb[u].z, bl[3] += vb[u].w;
      }
    }
    for (; k < K; ++k) {
      const long s = (long)k * N + p;
      const float4 vr = *reinterpret_cast<const float4*>(final_rgb + s);
      const float4 vg = *reinterpret_cast<const float4*>(final_rgb + FS + s);
      const float4 vb = *reinterpret_cast<const float4*>(final_rgb + 2 * FS + s);
      r[0] += vr.x, r[1] += vr.y, r[2] += vr.z, r[3] += vr.w;
      g[0] += vg.x, g[1] += vg.y, g[2] += vg.z, g[3] += vg.w;
      bl[0] += vb.x, bl[1] += vb.y, bl[2] += vb.z, bl[3] += vb.w;
    }
    img[0] = make_float4(r[0], g[0], bl[0], r[1]);
    img[1] = make_float4(g[1], bl[1], r[2], g[2]);
    img[2] = make_float4(bl[2], r[3], g[3], bl[3]);
  }
}
// scattered 4-byte writes into final (like retirement), to leave the buffer in the state the real gather finds it
__global__ __launch_bounds__(256) void kscatter(float* __restrict__ fin, long total) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long j = (i * 2654435761L) % total;
    fin[j] = 1.0f;
  }
}
int main() {
  const int N = 1920 * 1080, K = 24;
  for (long pad : {0L, 64L, 256L, 1056L, 4160L}) for (int grid : {2048, 4096, 8192}) {
    const long Np = N + pad;
    float *fin, *img;
    CK(hipMalloc(&fin, 3L * K * Np * 4 + 4096)); CK(hipMalloc(&img, (long)N * 4));
    CK(hipMemset(fin, 0, 3L * K * Np * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, fin, img, N / 4, Np, K, 4);
    CK(hipEventRecord(e0));
    const int reps = 10;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, fin, img, N / 4, Np, K, 4);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("pad %5ld floats, grid %5d: %.1f us/launch, %.2f TB/s\n", pad, grid, ms * 1e3 / reps, 3.0 * K * N * 4 * reps / (ms * 1e-3) / 1e12);
    CK(hipFree(fin)); CK(hipFree(img));
  }
  {
    const int K2 = 25; float *fin, *img;
    CK(hipMalloc(&fin, 3L * K2 * N * 4)); CK(hipMalloc(&img, 3L * N * 4));
    CK(hipMemset(fin, 0, 3L * K2 * N * 4)); CK(hipMemset(img, 0, 3L * N * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; ++mode) {
      float tot = 0;
      for (int r = 0; r < 6; ++r) {
        if (mode) hipLaunchKernelGGL(kscatter, dim3(4096), dim3(256), 0, 0, fin, 3L * K2 * N);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kg4, dim3(2025), dim3(256), 0, 0, fin, img, N, K2);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r) tot += ms;
      }
      printf("k_gather4 K=25 %s: %.1f us/launch, %.2f TB/s\n", mode ? "after scattered writes" : "clean", tot * 1e3 / 5, (3.0 * K2 + 6) * N * 4 / (tot / 5 * 1e-3) / 1e12);
    }
  }
  return 0;
}
