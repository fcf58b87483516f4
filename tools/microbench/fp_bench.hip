// Throughput of the Fp primitives of plonky2-bls12-381-pairing_amd/csrc/fp.hip.h on gfx950, per SIMD and
// chip-wide, at 1/2/4 waves per SIMD.  Feeds DESIGN.md's integer roofline for the pairing kernels.
//
// build: hipcc -O3 --offload-arch=gfx950 -I../../plonky2-bls12-381-pairing_amd/csrc fp_bench.hip -o fp_bench
#include "fp.hip.h"
#include <cstdio>
#include <vector>
using namespace bls;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

constexpr int ITERS = 256;

enum { OP_MUL, OP_MUL2, OP_SQR, OP_ADDSUB, OP_FP2MUL, OP_MUL2_PINNED, N_OPS };

// mul2 with every MAD pinned into ONE accumulator chain per column by inline asm (no compiler re-association,
// hence no 64-bit merge adds); tests whether a strictly sequential chain is latency-bound at low occupancy.
__device__ __forceinline__ void mad_pin(uint64_t& acc, uint32_t a, uint32_t b) {
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc");
}
__device__ __forceinline__ FpR mul2_pinned(const FpR& a, const FpR& b, const FpR& c, const FpR& d) {
  uint32_t m[NL]; FpR out; uint64_t acc = 0;
  uint32_t P[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) P[i] = BLS_P28[i];
#pragma unroll
  for (int k = 0; k < NL; ++k) {
#pragma unroll
    for (int i = 0; i <= k; ++i) { mad_pin(acc, a.l[i], b.l[k - i]); mad_pin(acc, c.l[i], d.l[k - i]); }
#pragma unroll
    for (int i = 0; i < k; ++i) mad_pin(acc, m[i], P[k - i]);
    m[k] = ((uint32_t)acc * BLS_PINV28) & LMASK;
    mad_pin(acc, m[k], P[0]);
    acc >>= LW;
  }
#pragma unroll
  for (int k = NL; k < 2 * NL - 1; ++k) {
#pragma unroll
    for (int i = k - NL + 1; i < NL; ++i) { mad_pin(acc, a.l[i], b.l[k - i]); mad_pin(acc, c.l[i], d.l[k - i]); }
#pragma unroll
    for (int i = k - NL + 1; i < NL; ++i) mad_pin(acc, m[i], P[k - i]);
    out.l[k - NL] = (uint32_t)acc & LMASK;
    acc >>= LW;
  }
  out.l[NL - 1] = (uint32_t)acc;
  return out;
}
static const char* names[] = {"mont mul (a*b)", "mont mul2 (a*b+c*d)", "mont sqr", "add+sub pair", "fp2 mul (2x mul2 + neg)", "mul2 pinned chain"};
static const double mads[] = {392, 588, 301, 0, 1176, 588};

template <int OP>
__global__ void __launch_bounds__(256) k_fp(const uint64_t* in, uint64_t* out, int iters) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  FpR x = fp_load(in + 6 * (t & 255), 0), y = fp_load(in + 6 * ((t + 1) & 255), 0);
  FpR z = fp_load(in + 6 * ((t + 2) & 255), 0), w = fp_load(in + 6 * ((t + 3) & 255), 0);
  for (int i = 0; i < iters; ++i) {
    if (OP == OP_MUL) { x = mul(x, y); y = mul(y, x); }
    else if (OP == OP_MUL2) { x = mul2(x, y, z, w); z = mul2(z, x, y, w); }
    else if (OP == OP_MUL2_PINNED) { x = mul2_pinned(x, y, z, w); z = mul2_pinned(z, x, y, w); }
    else if (OP == OP_SQR) { x = sqr(x); y = sqr(y); }
    else if (OP == OP_ADDSUB) {
#pragma unroll
      for (int r = 0; r < 8; ++r) { auto s = add(x, y); auto d = sub(s, z); x = assume_bounds<1, 2>(norm(d)); auto s2 = add(y, x); auto d2 = sub(s2, w); y = assume_bounds<1, 2>(norm(d2)); }
    } else if (OP == OP_FP2MUL) {
      // (x + y u)(z + w u) = (xz - yw) + (xw + yz) u, twice per trip
      FpR c0 = mul2(x, z, neg(y), w), c1 = mul2(x, w, y, z); x = c0; y = c1;
      c0 = mul2(x, z, neg(y), w); c1 = mul2(x, w, y, z); z = c0; w = c1;
    }
  }
  auto s = add(add(x, y), add(z, w));
  fp_store(out + 6 * t, s, 0);
}

template <int OP>
int run(const uint64_t* d_in, uint64_t* d_out, int n_cu) {
  for (int wps : {1, 2, 4}) {
    int blocks = n_cu * wps;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k_fp<OP><<<blocks, 256>>>(d_in, d_out, 8);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
      CK(hipEventRecord(e0));
      k_fp<OP><<<blocks, 256>>>(d_in, d_out, ITERS);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    double ops_per_lane = (OP == OP_ADDSUB) ? ITERS * 16.0 : ITERS * 2.0;   // ops of the named kind
    double ns_per_op_simd = best * 1e6 / (ops_per_lane * wps);                // per wave-op per SIMD
    double chip = ops_per_lane * 256.0 * blocks / (best * 1e-3);
    printf("%-26s waves/SIMD=%d  %8.3f ms  ns/op/SIMD=%9.2f  cyc@2.4GHz=%8.1f  cyc/MAD=%5.2f  chip=%8.2f Gop/s\n",
           names[OP], wps, best, ns_per_op_simd, ns_per_op_simd * 2.4,
           mads[OP] > 0 ? ns_per_op_simd * 2.4 / mads[OP] : 0.0, chip * 1e-9);
  }
  return 0;
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  int n_cu = p.multiProcessorCount;
  std::vector<uint64_t> h(256 * 6);
  uint64_t s = 0x424C533132333831ull;
  for (auto& v : h) { s = s * 6364136223846793005ull + 1442695040888963407ull; v = s; }
  for (int i = 0; i < 256; ++i) h[6 * i + 5] &= 0x0fffffffffffffffull;   // < p
  uint64_t *d_in, *d_out;
  CK(hipMalloc(&d_in, h.size() * 8)); CK(hipMalloc(&d_out, (size_t)n_cu * 4 * 256 * 48));
  CK(hipMemcpy(d_in, h.data(), h.size() * 8, hipMemcpyHostToDevice));
  if (run<OP_MUL>(d_in, d_out, n_cu)) return 1;
  if (run<OP_MUL2>(d_in, d_out, n_cu)) return 1;
  if (run<OP_SQR>(d_in, d_out, n_cu)) return 1;
  if (run<OP_FP2MUL>(d_in, d_out, n_cu)) return 1;
  if (run<OP_ADDSUB>(d_in, d_out, n_cu)) return 1;
  if (run<OP_MUL2_PINNED>(d_in, d_out, n_cu)) return 1;
  return 0;
}
