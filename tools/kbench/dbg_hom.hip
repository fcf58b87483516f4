// Developer tool: inline vs out-of-line homogeneous Miller steps on the same inputs (hunting a call-path mismatch).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "pairing.hip.h"
using namespace bls;
BLS_HD_NOINLINE LineCoeffs addition_step_hom_call(G2Jac& r, const G2Aff& q) { return addition_step_hom(r, q); }
__global__ void k(const uint64_t* g2, uint64_t* out) {
  const size_t i = threadIdx.x / 2;
  bool inf;
  G2Aff q = g2_aff_load(g2 + 24 * i, 0, inf);
  G2Jac r1; r1.x = q.x; r1.y = q.y; r1.z = Fp2R(fp2_one());
  (void)doubling_step_hom(r1);
  G2Jac r2 = r1;
  jac_store(out + 36 * (3 * i + 0), r1, 0);
  (void)addition_step_hom(r1, q);
  (void)addition_step_hom_call(r2, q);
  jac_store(out + 36 * (3 * i + 1), r1, 0);
  jac_store(out + 36 * (3 * i + 2), r2, 0);
}
int main() {
  std::vector<uint64_t> h(24 * 32);
  uint64_t s = 12345; for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = s; } for (size_t i = 5; i < h.size(); i += 6) h[i] &= 0x0fffffffffffffffull;
  uint64_t* d; uint64_t* o; hipMalloc(&d, h.size() * 8); hipMalloc(&o, 32 * 3 * 36 * 8);
  hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  k<<<1, 64>>>(d, o); hipDeviceSynchronize();
  std::vector<uint64_t> r(3 * 36); hipMemcpy(r.data(), o, r.size() * 8, hipMemcpyDeviceToHost);
  printf("Q"); for (int j = 0; j < 24; ++j) printf(" %016llx", (unsigned long long)h[j]); printf("\n");
  const char* nm[3] = {"R_after_dbl", "R_add_inline", "R_add_call"};
  for (int t = 0; t < 3; ++t) { printf("%s", nm[t]); for (int j = 0; j < 36; ++j) printf(" %016llx", (unsigned long long)r[36 * t + j]); printf("\n"); }
  return 0;
}
