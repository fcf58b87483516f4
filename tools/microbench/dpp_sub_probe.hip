// Hardware probe (gfx950): what VOP2 "rev" opcodes compute in their DPP encoding.  LLVM's DPP combiner (GCNDPPCombine)
// folds   o = v_mov_b32_dpp(a);  d = b - o   into   v_subrev_u32_dpp d, a, b   assuming  d = b - dpp(a).
// Finding (gpurun_out/dpp_sub_probe.txt): the hardware returns dpp(a) - b, i.e. the DPP forms of v_sub_u32 and v_subrev_u32
// compute the same thing.  Root cause of the wrong Fp2 squarings described in DESIGN 3.3.
#include <hip/hip_runtime.h>
#include <cstdio>
#define DPP " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1"
__global__ void k(int* out) {
  const int l = threadIdx.x;
  int a = 100 * (l + 1);                  // lane l: 100 (l + 1); partner lane l ^ 1
  int b = 7 * (l + 1) + 1000;             // lane l: 7 (l + 1) + 1000
  int sh = 1 + (l & 1);                   // shift amounts: even lanes 1, odd lanes 2
  float fa = (float)a, fb = (float)b;
  int r[12];
  asm volatile("v_sub_u32_dpp %0, %1, %2" DPP : "=&v"(r[0]) : "v"(a), "v"(b));
  asm volatile("v_subrev_u32_dpp %0, %1, %2" DPP : "=&v"(r[1]) : "v"(a), "v"(b));
  asm volatile("v_sub_u32 %0, %1, %2" : "=&v"(r[2]) : "v"(a), "v"(b));
  asm volatile("v_subrev_u32 %0, %1, %2" : "=&v"(r[3]) : "v"(a), "v"(b));
  asm volatile("v_lshlrev_b32_dpp %0, %1, %2" DPP : "=&v"(r[4]) : "v"(sh), "v"(b));
  asm volatile("v_lshrrev_b32_dpp %0, %1, %2" DPP : "=&v"(r[5]) : "v"(sh), "v"(b));
  float f0, f1;
  asm volatile("v_sub_f32_dpp %0, %1, %2" DPP : "=&v"(f0) : "v"(fa), "v"(fb));
  asm volatile("v_subrev_f32_dpp %0, %1, %2" DPP : "=&v"(f1) : "v"(fa), "v"(fb));
  r[6] = (int)f0; r[7] = (int)f1;
  asm volatile("v_subrev_co_u32_dpp %0, vcc, %1, %2" DPP : "=&v"(r[8]) : "v"(a), "v"(b) : "vcc");
  asm volatile("v_add_u32_dpp %0, %1, %2" DPP : "=&v"(r[9]) : "v"(a), "v"(b));
  asm volatile("v_mov_b32_dpp %0, %1" DPP : "=&v"(r[10]) : "v"(a));
  r[11] = 0;
  for (int j = 0; j < 12; ++j) out[l * 16 + j] = r[j];
  out[l * 16 + 12] = a; out[l * 16 + 13] = b; out[l * 16 + 14] = sh;
}
int main() {
  int* d; (void)hipMalloc(&d, 64 * 16 * 4);
  k<<<1, 64>>>(d); (void)hipDeviceSynchronize();
  int h[64 * 16]; (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  const char* nm[11] = {"v_sub_u32_dpp a,b", "v_subrev_u32_dpp a,b", "v_sub_u32 a,b (no dpp)", "v_subrev_u32 a,b (no dpp)", "v_lshlrev_b32_dpp sh,b",
                        "v_lshrrev_b32_dpp sh,b", "v_sub_f32_dpp a,b", "v_subrev_f32_dpp a,b", "v_subrev_co_u32_dpp a,b", "v_add_u32_dpp a,b", "v_mov_b32_dpp a"};
  for (int l = 0; l < 2; ++l) {
    const int a = h[l * 16 + 12], b = h[l * 16 + 13], sh = h[l * 16 + 14], pa = h[(l ^ 1) * 16 + 12], pb = h[(l ^ 1) * 16 + 13], psh = h[(l ^ 1) * 16 + 14];
    printf("lane %d: a=%d b=%d sh=%d | partner a=%d b=%d sh=%d\n", l, a, b, sh, pa, pb, psh);
    for (int j = 0; j < 11; ++j) {
      const int v = h[l * 16 + j];
      printf("  %-28s = %-8d", nm[j], v);
      if (v == pa - b) printf(" [dpp(S0) - S1]");
      if (v == b - pa) printf(" [S1 - dpp(S0)]");
      if (v == a - b) printf(" [S0 - S1]");
      if (v == b - a) printf(" [S1 - S0]");
      if (v == pb - a) printf(" [dpp(S1) - S0]");
      if (v == a - pb) printf(" [S0 - dpp(S1)]");
      if (v == (b << psh)) printf(" [S1 << dpp(S0)]");
      if (v == (b >> psh)) printf(" [S1 >> dpp(S0)]");
      if (v == (pb << sh)) printf(" [dpp(S1) << S0]");
      if (v == (psh << b)) printf(" [dpp(S0) << S1]");
      if (v == pa + b) printf(" [dpp(S0) + S1]");
      printf("\n");
    }
  }
  return 0;
}
