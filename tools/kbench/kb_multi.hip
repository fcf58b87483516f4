// Developer tool: the Miller kernel for products of K pairs, register/scratch form (miller_ate) against the LDS form
// (miller_ate_lds), same inputs, checksums must agree.   KB_K = 2..3, KB_HOM = true/false
// (the P-in-LDS / Q-re-read variants of round 3 are in the git history of csrc/pairing.hip.h: DESIGN 3.8 has their numbers)
//   KB_SRC=kb_multi.hip tools/kbench/build_variant.sh multi3 -DKB_K=3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define BLS_LAZY_014 1
#include "pairing.hip.h"
using namespace bls;
#ifndef KB_K
#define KB_K 3
#endif
#ifndef KB_HOM
#define KB_HOM true
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

__global__ void __launch_bounds__(256)
k_old(const uint64_t* __restrict__ g1, const uint64_t* __restrict__ g2, uint64_t* __restrict__ out, size_t n) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / LANES_PER_ELEM;
  if (i >= n) return;
  G1Aff p[KB_K]; G2Aff q[KB_K]; bool skip[KB_K];
#pragma unroll
  for (int j = 0; j < KB_K; ++j) {
    bool i1, i2;
    p[j] = g1_aff_load(g1 + 12 * (KB_K * i + j), 0, i1);
    q[j] = g2_aff_load(g2 + 24 * (KB_K * i + j), 0, i2);
    skip[j] = i1 || i2;
  }
  fp12_store(out + 72 * i, miller_ate<KB_K, KB_HOM>(p, q, skip), 0);
}
__global__ void __launch_bounds__(64)
k_lds(const uint64_t* __restrict__ g1, const uint64_t* __restrict__ g2, uint64_t* __restrict__ out, size_t n) {
  extern __shared__ LdsQuad lds_mem[];
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / LANES_PER_ELEM;
  if (i >= n) return;
  uint8_t fl = 0;
  const LaneLds<64> lds = lane_lds<64>(lds_mem, threadIdx.x);
  fp12_store(out + 72 * i, miller_ate_lds<KB_K, KB_HOM>(g1 + 12 * KB_K * i, g2 + 24 * KB_K * i, 0, lds, fl), 0);
}

int main(int argc, char** argv) {
  int lg = argc > 1 ? atoi(argv[1]) : 18;
  int block = argc > 2 ? atoi(argv[2]) : 64;
  size_t n = (size_t)1 << lg;
  const size_t np = n * KB_K;
  std::vector<uint64_t> h1(12 * 192), h2(24 * 192);
  uint64_t s = 0x424C533132333831ull;
  auto nxt = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
  for (size_t i = 0; i < h1.size(); ++i) { h1[i] = nxt(); if (i % 6 == 5) h1[i] &= 0x0fffffffffffffffull; }
  for (size_t i = 0; i < h2.size(); ++i) { h2[i] = nxt(); if (i % 6 == 5) h2[i] &= 0x0fffffffffffffffull; }
  uint64_t *d1, *d2, *dout;
  CK(hipMalloc(&d1, (np + 192) * 96)); CK(hipMalloc(&d2, (np + 192) * 192)); CK(hipMalloc(&dout, n * 576));
  for (size_t i = 0; i < np; i += 192) {
    CK(hipMemcpy(d1 + 12 * i, h1.data(), 96 * 192, hipMemcpyHostToDevice));
    CK(hipMemcpy(d2 + 24 * i, h2.data(), 192 * 192, hipMemcpyHostToDevice));
  }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  unsigned grid = (unsigned)((n * LANES_PER_ELEM + block - 1) / block);
  const size_t shm = (size_t)block * miller_lds_bytes_per_lane(KB_K);
  uint64_t sums[2];
  for (int v = 0; v < 2; ++v) {
    CK(hipMemset(dout, 0, n * 576));
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
      CK(hipEventRecord(e0));
      if (v == 0) k_old<<<grid, block>>>(d1, d2, dout, n);
      else k_lds<<<grid, block, shm>>>(d1, d2, dout, n);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      CK(hipGetLastError());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r && ms < best) best = ms;
    }
    std::vector<uint64_t> ho(72 * 192);
    CK(hipMemcpy(ho.data(), dout + 72 * (n - 192), 576 * 192, hipMemcpyDeviceToHost));
    uint64_t x = 0; for (auto w : ho) x = (x * 1099511628211ull) ^ w;
    sums[v] = x;
    hipFuncAttributes fa; CK(hipFuncGetAttributes(&fa, v == 0 ? (const void*)k_old : (const void*)k_lds));
    printf("%-10s K=%d hom=%d %-4s n=2^%d block=%d  %9.3f ms  vgpr=%d scratch=%zu B lds=%zu B  checksum=%016llx\n",
           KB_NAME, KB_K, (int)KB_HOM, v == 0 ? "old" : "lds", lg, block, best, fa.numRegs, (size_t)fa.localSizeBytes, v ? shm : (size_t)0, (unsigned long long)x);
  }
  printf("%s\n", sums[0] == sums[1] ? "checksums agree" : "CHECKSUMS DIFFER");
  return sums[0] == sums[1] ? 0 : 2;
}
