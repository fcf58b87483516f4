// Instruction-issue microbenchmark for gfx950: measures the per-SIMD cost of the
// VALU instructions a 381-bit Montgomery multiplier can be built from, at 1/2/4/8
// waves per SIMD.  Output feeds DESIGN.md's integer roofline ("peak limb-MAC/s").
//
// build: hipcc -O3 --offload-arch=gfx950 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

constexpr int ITERS = 2048;     // loop trips
constexpr int CHAINS = 8;       // independent dependency chains per lane
constexpr int UNROLL = 4;       // ops per chain per trip

enum Op { MAD_U64_U32, MUL_LO_U32, MUL_HI_U32, MAD_U32_U24, MUL_HI_U32_U24, ADD_CO, ADDC_CO,
          ADD3_U32, FMA_F64, FMA_F32, LSHRREV_B64, ADD_U32, MAD_U64_DEP, N_OPS };
static const char* op_names[] = {"v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u32_u24",
  "v_mul_hi_u32_u24", "v_add_co_u32", "v_addc_co_u32", "v_add3_u32", "v_fma_f64", "v_fma_f32",
  "v_lshrrev_b64", "v_add_u32", "v_mad_u64_u32(dep-chain x1)"};

template <int OP>
__global__ void __launch_bounds__(256) k_rate(uint32_t* out, uint32_t seed) {
  uint32_t t = threadIdx.x + blockIdx.x * blockDim.x;
  uint64_t acc[CHAINS];
  double dacc[CHAINS];
  float facc[CHAINS];
  uint32_t a = seed * 2654435761u + t, b = (seed ^ 0x9e3779b9u) + t * 7u;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) { acc[c] = (uint64_t)(t + c) * 0x9e3779b97f4a7c15ull; dacc[c] = 1.0 + c + t; facc[c] = 1.0f + c; }
  double da = 1.0000001 + (double)(t & 3) * 1e-9, db = 1e-9;
  float fa = 1.0000001f, fb = 1e-7f;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      if (OP == MAD_U64_DEP) {
        // single dependent chain: latency
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[0]) : "v"(a), "v"(b) : "vcc");
        continue;
      }
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) {
        uint32_t lo = (uint32_t)acc[c], hi = (uint32_t)(acc[c] >> 32);
        if (OP == MAD_U64_U32) {
          asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(b) : "vcc");
        } else if (OP == MUL_LO_U32) {
          asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(lo) : "v"(a)); acc[c] = lo | ((uint64_t)hi << 32);
        } else if (OP == MUL_HI_U32) {
          asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(lo) : "v"(a)); acc[c] = lo | ((uint64_t)hi << 32);
        } else if (OP == MAD_U32_U24) {
          asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(lo) : "v"(a), "v"(b)); acc[c] = lo | ((uint64_t)hi << 32);
        } else if (OP == MUL_HI_U32_U24) {
          asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(lo) : "v"(a)); acc[c] = lo | ((uint64_t)hi << 32);
        } else if (OP == ADD_CO) {
          asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(lo) : "v"(a) : "vcc"); acc[c] = lo | ((uint64_t)hi << 32);
        } else if (OP == ADDC_CO) {
          asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(lo) : "v"(a) : "vcc"); acc[c] = lo | ((uint64_t)hi << 32);
        } else if (OP == ADD3_U32) {
          asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(lo) : "v"(a), "v"(b)); acc[c] = lo | ((uint64_t)hi << 32);
        } else if (OP == ADD_U32) {
          asm volatile("v_add_u32 %0, %0, %1" : "+v"(lo) : "v"(a)); acc[c] = lo | ((uint64_t)hi << 32);
        } else if (OP == LSHRREV_B64) {
          asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(acc[c]));
        } else if (OP == FMA_F64) {
          asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(dacc[c]) : "v"(da), "v"(db));
        } else if (OP == FMA_F32) {
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(facc[c]) : "v"(fa), "v"(fb));
        }
      }
    }
  }
  uint64_t s = 0; double ds = 0; float fs = 0;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) { s ^= acc[c]; ds += dacc[c]; fs += facc[c]; }
  out[t] = (uint32_t)s ^ (uint32_t)(s >> 32) ^ (uint32_t)ds ^ (uint32_t)fs;
}

// clock probe: ratio of shader clock (s_memtime) to 100 MHz realtime under an integer-multiply load
__global__ void __launch_bounds__(256) k_clock(uint64_t* out, uint32_t seed) {
  uint32_t t = threadIdx.x + blockIdx.x * blockDim.x;
  uint64_t acc[CHAINS];
  uint32_t a = seed * 2654435761u + t, b = (seed ^ 0x9e3779b9u) + t * 7u;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) acc[c] = (uint64_t)(t + c) * 0x9e3779b97f4a7c15ull;
  uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < ITERS * 8; ++it) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(b) : "vcc");
  }
  uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  uint64_t s = 0;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) s ^= acc[c];
  if (threadIdx.x == 0) { out[blockIdx.x * 3 + 0] = t1 - t0; out[blockIdx.x * 3 + 1] = r1 - r0; out[blockIdx.x * 3 + 2] = s; }
}

template <int OP>
int run(uint32_t* d_out, int n_cu) {
  for (int wps : {1, 2, 4, 8}) {              // waves per SIMD (256-thread block = 1 wave on each of 4 SIMDs)
    int blocks = n_cu * wps;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k_rate<OP><<<blocks, 256>>>(d_out, 1);    // warm
    CK(hipDeviceSynchronize());
    const int reps = 5; float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
      CK(hipEventRecord(e0));
      k_rate<OP><<<blocks, 256>>>(d_out, r + 2);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    double per_chain = (OP == MAD_U64_DEP) ? 1.0 : (double)CHAINS;
    double wave_instr_per_simd = (double)ITERS * UNROLL * per_chain * wps;   // per SIMD
    double ns_per_instr = best * 1e6 / wave_instr_per_simd;
    double lane_ops = wave_instr_per_simd * 64.0 * 4.0 * n_cu;
    printf("%-28s waves/SIMD=%d  time=%8.3f ms  ns/wave-instr/SIMD=%7.3f  (cyc@2.4GHz=%6.2f)  chip=%9.1f Gop/s\n",
           op_names[OP], wps, best, ns_per_instr, ns_per_instr * 2.4, lane_ops / (best * 1e6));
  }
  return 0;
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  int n_cu = p.multiProcessorCount;
  printf("device=%s arch=%s CUs=%d clock=%d kHz\n", p.name, p.gcnArchName, n_cu, p.clockRate);
  uint32_t* d_out; CK(hipMalloc(&d_out, (size_t)n_cu * 8 * 256 * 4 * 2));
  if (run<MAD_U64_U32>(d_out, n_cu)) return 1;
  if (run<MAD_U64_DEP>(d_out, n_cu)) return 1;
  if (run<MUL_LO_U32>(d_out, n_cu)) return 1;
  if (run<MUL_HI_U32>(d_out, n_cu)) return 1;
  if (run<MAD_U32_U24>(d_out, n_cu)) return 1;
  if (run<MUL_HI_U32_U24>(d_out, n_cu)) return 1;
  if (run<ADD_CO>(d_out, n_cu)) return 1;
  if (run<ADDC_CO>(d_out, n_cu)) return 1;
  if (run<ADD3_U32>(d_out, n_cu)) return 1;
  if (run<ADD_U32>(d_out, n_cu)) return 1;
  if (run<LSHRREV_B64>(d_out, n_cu)) return 1;
  if (run<FMA_F64>(d_out, n_cu)) return 1;
  if (run<FMA_F32>(d_out, n_cu)) return 1;
  // clock under load
  uint64_t* d_clk; CK(hipMalloc(&d_clk, (size_t)n_cu * 4 * 3 * 8));
  for (int i = 0; i < 3; ++i) { k_clock<<<n_cu * 4, 256>>>(d_clk, i); }
  CK(hipDeviceSynchronize());
  std::vector<uint64_t> h(n_cu * 4 * 3);
  CK(hipMemcpy(h.data(), d_clk, h.size() * 8, hipMemcpyDeviceToHost));
  double sum = 0; for (int b = 0; b < n_cu * 4; ++b) sum += (double)h[b * 3] / (double)h[b * 3 + 1] * 100.0;
  printf("in-kernel clock under v_mad_u64_u32 load (4 waves/SIMD): %.1f MHz (mean over blocks)\n", sum / (n_cu * 4));
  return 0;
}
