// Developer tool: times the pairing kernel under different compile-time structure flags (inlining granularity,
// launch bounds) without going through Python.  Inputs are random field elements (timing is data independent);
// the XOR checksum of all outputs must be identical across variants.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../plonky2-bls12-381-pairing_amd/csrc [-D...] kbench.hip -o kbench_X
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#ifdef KB_SYNC
#define BLS_MILLER_ITERATION_HOOK() __syncthreads()
#endif
#ifdef KB_HOOK
#define BLS_MILLER_ITERATION_HOOK() KB_HOOK
#endif
#include "pairing.hip.h"
using namespace bls;
#ifndef KB_LB
#define KB_LB 256
#endif
#ifndef KB_ATTR
#define KB_ATTR
#endif
#ifndef KB_FE
#define KB_FE 1
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

__global__ void __launch_bounds__(KB_LB) KB_ATTR
k_pair(const uint64_t* __restrict__ g1, const uint64_t* __restrict__ g2, uint64_t* __restrict__ out, size_t n) {
  size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / LANES_PER_ELEM;
#if defined(KB_SYNC) || defined(KB_CLAMP)
  if (i >= n) i = n - 1;             // every lane reaches the barriers (n is a multiple of the block in this tool)
#else
  if (i >= n) return;
#endif
  G1Aff p[1]; G2Aff q[1]; bool skip[1];
  bool i1, i2;
  p[0] = g1_aff_load(g1 + 12 * i, 0, i1);
  q[0] = g2_aff_load(g2 + 24 * i, 0, i2);
  skip[0] = i1 || i2;
#ifdef KB_ONLY_FE
  Fp12R f = fp12_load(g2 + 24 * i, 0);      // any 12 field elements (n must be <= 2^18 / 3 of the g2 buffer: use lg <= 16)
  (void)p; (void)q; (void)skip;
  f = final_exponentiation(f);
#else
#ifndef KB_HOM
#define KB_HOM false
#endif
  Fp12R f = miller_ate<1, KB_HOM>(p, q, skip);
  if (KB_FE) f = final_exponentiation(f);
#endif
  fp12_store(out + 72 * i, f, 0);
}

int main(int argc, char** argv) {
  int lg = argc > 1 ? atoi(argv[1]) : 18;
  int block = argc > 2 ? atoi(argv[2]) : 64;
  size_t n = (size_t)1 << lg;
  std::vector<uint64_t> h1(12 * 64), h2(24 * 64);
  uint64_t s = 0x424C533132333831ull;
  auto nxt = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
  for (size_t i = 0; i < h1.size(); ++i) { h1[i] = nxt(); if (i % 6 == 5) h1[i] &= 0x0fffffffffffffffull; }
  for (size_t i = 0; i < h2.size(); ++i) { h2[i] = nxt(); if (i % 6 == 5) h2[i] &= 0x0fffffffffffffffull; }
  uint64_t *d1, *d2, *dout;
  CK(hipMalloc(&d1, n * 96)); CK(hipMalloc(&d2, n * 192 * 3 + 4096)); CK(hipMalloc(&dout, n * 576));
  for (size_t i = 0; i < n; i += 64) {
    CK(hipMemcpy(d1 + 12 * i, h1.data(), 96 * 64, hipMemcpyHostToDevice));
    CK(hipMemcpy(d2 + 24 * i, h2.data(), 192 * 64, hipMemcpyHostToDevice));
  }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  unsigned grid = (unsigned)((n * LANES_PER_ELEM + block - 1) / block);
  k_pair<<<grid, block>>>(d1, d2, dout, n);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 2; ++r) {
    CK(hipEventRecord(e0));
    k_pair<<<grid, block>>>(d1, d2, dout, n);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  std::vector<uint64_t> ho(72 * 64);
  CK(hipMemcpy(ho.data(), dout + 72 * (n - 64), 576 * 64, hipMemcpyDeviceToHost));
  uint64_t x = 0; for (auto v : ho) x = (x * 1099511628211ull) ^ v;
  hipFuncAttributes fa; CK(hipFuncGetAttributes(&fa, (const void*)k_pair));
  printf("%-40s n=2^%d block=%d  %9.3f ms  %8.1f k pairings/s  vgpr=%d scratch=%zu B  checksum=%016llx\n",
         KB_NAME, lg, block, best, n / best, fa.numRegs, (size_t)fa.localSizeBytes, (unsigned long long)x);
  return 0;
}
