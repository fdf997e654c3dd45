// Micro-benchmark: persistent-grid streaming of path records with different layouts.
//  mode 0: 6 dword plane loads + 8 dword plane stores per lane (SoA of scalars, current layout)
//  mode 1: float3 + float3 loads, float4 + float4 stores (SoA of small vectors)
//  mode 2: float4 + float2 loads, float4 + float4 stores
// All move 56 B per element.  Reports TB/s for N elements.
#include <hip/hip_runtime.h>
#include <cstdio>
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
int main() {
  for (long n : {12L << 20, 3L << 20}) for (int bpc : {5, 8}) {
    if (run<0>("SoA dword (6 ld + 8 st)", n, bpc)) return 1;
    if (run<1>("float3,float3 -> float4,float4", n, bpc)) return 1;
    if (run<2>("float4,float2 -> float4,float4", n, bpc)) return 1;
  }
  return 0;
}
