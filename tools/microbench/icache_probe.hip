// Does straight-line code larger than the instruction cache slow a single wave per SIMD down?
// Body = UNR x 8 independent v_mad_u64_u32 (8 bytes each); total MADs executed is constant across variants.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s\n", hipGetErrorString(e_)); return 1; } } while (0)
template <int UNR>
__global__ void __launch_bounds__(256) k(uint32_t* out, uint32_t seed, int trips) {
  uint32_t t = threadIdx.x + blockIdx.x * blockDim.x;
  uint64_t acc[8];
  uint32_t a = seed * 2654435761u + t, b = (seed ^ 0x9e3779b9u) + t * 7u;
#pragma unroll
  for (int c = 0; c < 8; ++c) acc[c] = (uint64_t)(t + c) * 0x9e3779b97f4a7c15ull;
  for (int it = 0; it < trips; ++it) {
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
#pragma unroll
      for (int c = 0; c < 8; ++c) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(b) : "vcc");
    }
  }
  uint64_t s = 0;
#pragma unroll
  for (int c = 0; c < 8; ++c) s ^= acc[c];
  out[t] = (uint32_t)s ^ (uint32_t)(s >> 32);
}
template <int UNR> int run(uint32_t* d, int n_cu) {
  const long total = 1 << 17;            // MAD groups per lane
  int trips = (int)(total / UNR);
  for (int wps : {1, 2}) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k<UNR><<<n_cu * wps, 256>>>(d, 1, 2); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) { CK(hipEventRecord(e0)); k<UNR><<<n_cu * wps, 256>>>(d, r + 2, trips); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; }
    double instr = (double)total * 8 * wps;   // MADs (+ about 1 VALU add each) per SIMD
    printf("body ~%4d KB  waves/SIMD=%d  %8.3f ms  ns per MAD-group-instr per SIMD = %.3f\n", UNR * 8 * 8 / 1024, wps, best, best * 1e6 / instr);
  }
  return 0;
}
int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  uint32_t* d; CK(hipMalloc(&d, (size_t)p.multiProcessorCount * 2 * 256 * 4));
  if (run<32>(d, p.multiProcessorCount)) return 1;
  if (run<256>(d, p.multiProcessorCount)) return 1;
  if (run<1024>(d, p.multiProcessorCount)) return 1;
  if (run<4096>(d, p.multiProcessorCount)) return 1;
  return 0;
}
