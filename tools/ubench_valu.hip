// Micro-benchmark: sustained wave64 VALU issue rate on gfx950 for the instruction mixes the
// intersect kernel uses, at 1..8 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  float x0 = threadIdx.x * 1e-3f + a, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f;
  float y0 = b, y1 = b + 1, y2 = b + 2, y3 = b + 3;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (MODE == 0) {  // 4 independent mul+add chains (8 VALU)
        x0 = x0 * a + b; x1 = x1 * a + b; x2 = x2 * a + b; x3 = x3 * a + b;
      } else if (MODE == 1) {  // one dependent chain (2 VALU)
        x0 = x0 * a + b;
      } else if (MODE == 2) {  // slab-like mix: cndmask, sub, mul, max (8 VALU)
        float lo = x0 < 0.f ? y0 : y1, hi = x0 < 0.f ? y1 : y0;
        float t0 = (lo - x1) * a, t1 = (hi - x1) * a;
        x2 = __builtin_fmaxf(x2, t0); x3 = __builtin_fminf(x3, t1);
        x0 = x0 + b;
      } else if (MODE == 3) {  // IEEE divide chain
        x0 = x0 / (y0 + x1); x1 = x1 + b;
      } else if (MODE == 4) {  // sqrt
        x0 = __builtin_sqrtf(x0 * x0 + b); 
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + y0 + y1 + y2 + y3;
}
template <int MODE>
void run(const char* name, int valu_per_iter) {
  float* d; hipMalloc(&d, 256 * 8 * 256 * 4 * 4);
  for (int bpc = 1; bpc <= 8; bpc *= 2) {
    int grid = 256 * bpc, iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, 10, 1.0001f, 1e-6f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, iters, 1.0001f, 1e-6f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double waves_per_simd = bpc;  // 256 thr = 4 waves = 1 per SIMD per block
    double instr_per_simd = (double)iters * 16 * valu_per_iter * waves_per_simd;
    printf("%-28s waves/SIMD %d: %.3f ms, %.2f ns per wave-instr per SIMD (%.2f cycles @2.4GHz)\n", name, bpc, ms,
           ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
  }
  hipFree(d);
}
int main() {
  run<0>("4 indep chains (packed, 4 VALU)", 4);
  run<1>("1 dependent mul+add chain", 2);
  run<2>("slab mix (8 VALU)", 9);
  run<3>("IEEE fp32 divide (+2)", 12);
  run<4>("IEEE sqrt (+2) = 18 VALU", 18);
  return 0;
}
