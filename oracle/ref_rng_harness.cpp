// oracle/ref_rng_harness.cpp — TEST INFRASTRUCTURE ONLY (golden generation).
//
// The reference draws its random numbers from CUDA Thrust (pathtrace.cu:9,204-206,368-369:
// thrust::default_random_engine seeded per (iter, index, depth), thrust::uniform_real_distribution<float> u01(0, 1)).
// CUDA Thrust is a third-party dependency that is NOT vendored under /root/reference (CUDA Toolkit >= 10 per
// CMakeLists.txt:26, version unpinned) and not installed here.  rocThrust 7.2 (/opt/rocm/include/thrust) ships the
// same published algorithm — minstd_rand (a = 48271, m = 2^31 - 1), seed 0 -> default seed, and
// uniform_real_distribution = (x - min) / (1 + float(max - min)) scaled to [a, b) — and compiles for the host with
// `hipcc -x hip --cuda-host-only`.  This harness runs THAT code on the seed list ref_hot_harness.cpp produced with the
// reference's own utilhash (section "seed_out" of ref_isect.bin) and records, per seed, five u01 draws (the most one
// shading call consumes: roulette, lobe choice, three specular draws) and the raw engine outputs of a second engine.
// It also records the default-constructed engine's 10000th output (the C++ standard's check value for minstd_rand).
//
//   ref_rng IN(ref_isect.bin) OUT(ref_rng.bin)
#include <thrust/random.h>

#include <cstdio>
#include <vector>

#include "ref_gold_io.h"

int main(int argc, char** argv) {
  if (argc < 3) {
    fprintf(stderr, "usage: ref_rng ref_isect.bin OUT.bin\n");
    return 2;
  }
  gold::File in;
  if (!in.read(argv[1])) return 1;
  const gold::Section* seeds = in.find("seed_out");
  if (!seeds) return 1;
  gold::File out;
  gold::Section& s = out.add("seed", 1);
  gold::Section& raw = out.add("raw", 5);
  gold::Section& u = out.add("u01", 5);
  for (uint32_t w : seeds->w) {
    int h = (int)w;
    thrust::default_random_engine a(h), b(h);  // constructed from an int, as pathtrace.cu:206 does
    thrust::uniform_real_distribution<float> u01(0, 1);
    s.w.push_back(w);
    for (int k = 0; k < 5; ++k) {
      raw.w.push_back((uint32_t)a());
      u.w.push_back(gold::fbits(u01(b)));
    }
  }
  gold::Section& chk = out.add("minstd_check", 4);  // 10000th output of the default engine, min, max, default seed's first
  thrust::minstd_rand d;
  uint32_t v = 0;
  for (int k = 0; k < 10000; ++k) v = (uint32_t)d();
  thrust::minstd_rand e;
  chk.w.push_back(v);
  chk.w.push_back((uint32_t)thrust::minstd_rand::min);
  chk.w.push_back((uint32_t)thrust::minstd_rand::max);
  chk.w.push_back((uint32_t)e());
  // extremes: seeds chosen (modular inverse of the multiplier, our arithmetic) so that the FIRST output is a given x:
  // the smallest and largest outputs and the band where float(x - 1) / 2^31 rounds up to exactly 1.0f (SURVEY a-8)
  gold::Section& ext = out.add("extreme", 3);  // seed, raw, u01 bits
  const unsigned long long M = 2147483647ull, A = 48271ull;
  unsigned long long inv = 1, base = A, e2 = M - 2;
  for (; e2; e2 >>= 1, base = base * base % M)
    if (e2 & 1) inv = inv * base % M;
  std::vector<uint32_t> sds = {0u, 2147483647u, 2147483648u, 0xffffffffu};
  for (unsigned long long x : {1ull, 2ull, M - 1, M - 2, M - 33, M - 64, M - 65, M - 66, M - 129, M - 200, M / 2})
    sds.push_back((uint32_t)(x * inv % M));
  for (uint32_t sd : sds) {
    thrust::default_random_engine a(sd), b(sd);
    thrust::uniform_real_distribution<float> u01(0, 1);
    ext.w.push_back(sd);
    ext.w.push_back((uint32_t)a());
    ext.w.push_back(gold::fbits(u01(b)));
  }
  return out.write(argv[2]) ? 0 : 1;
}
