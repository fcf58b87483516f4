// Host <-> device copy rates on the GPU box as the host-pointer entry points see them (DESIGN.md 4, item "from host"):
// pageable hipMemcpyAsync, pinned (hipHostMalloc) hipMemcpyAsync, hipHostRegister of an existing pageable buffer, and a
// CPU memcpy into a pinned staging buffer.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t in_b = (size_t)288 << 20, out_b = (size_t)576 << 20;          // 2^20 pairings: 288 B in, 576 B out each
  void *din, *dout; (void)hipMalloc(&din, in_b); (void)hipMalloc(&dout, out_b);
  hipStream_t s; (void)hipStreamCreate(&s);
  char* pin = (char*)malloc(in_b); char* pout = (char*)malloc(out_b);
  memset(pin, 1, in_b); memset(pout, 2, out_b);
  for (int rep = 0; rep < 3; ++rep) {
    double t0 = now(); (void)hipMemcpyAsync(din, pin, in_b, hipMemcpyHostToDevice, s); (void)hipStreamSynchronize(s); double t1 = now();
    (void)hipMemcpyAsync(pout, dout, out_b, hipMemcpyDeviceToHost, s); (void)hipStreamSynchronize(s); double t2 = now();
    printf("pageable  H2D %.1f ms (%.1f GB/s)   D2H %.1f ms (%.1f GB/s)\n", (t1 - t0) * 1e3, in_b / (t1 - t0) / 1e9, (t2 - t1) * 1e3, out_b / (t2 - t1) / 1e9);
  }
  char *hin, *hout;
  double a0 = now(); (void)hipHostMalloc((void**)&hin, in_b, hipHostMallocDefault); (void)hipHostMalloc((void**)&hout, out_b, hipHostMallocDefault); double a1 = now();
  printf("hipHostMalloc of %zu MB: %.1f ms\n", (in_b + out_b) >> 20, (a1 - a0) * 1e3);
  for (int rep = 0; rep < 3; ++rep) {
    double c0 = now(); memcpy(hin, pin, in_b); double c1 = now();
    double t0 = now(); (void)hipMemcpyAsync(din, hin, in_b, hipMemcpyHostToDevice, s); (void)hipStreamSynchronize(s); double t1 = now();
    (void)hipMemcpyAsync(hout, dout, out_b, hipMemcpyDeviceToHost, s); (void)hipStreamSynchronize(s); double t2 = now();
    double c2 = now(); memcpy(pout, hout, out_b); double c3 = now();
    printf("pinned    H2D %.1f ms (%.1f GB/s)   D2H %.1f ms (%.1f GB/s)   cpu memcpy in %.1f ms (%.1f GB/s) out %.1f ms (%.1f GB/s)\n", (t1 - t0) * 1e3,
           in_b / (t1 - t0) / 1e9, (t2 - t1) * 1e3, out_b / (t2 - t1) / 1e9, (c1 - c0) * 1e3, in_b / (c1 - c0) / 1e9, (c3 - c2) * 1e3, out_b / (c3 - c2) / 1e9);
  }
  for (int rep = 0; rep < 2; ++rep) {
    double r0 = now(); hipError_t e1 = hipHostRegister(pin, in_b, hipHostRegisterDefault); hipError_t e2 = hipHostRegister(pout, out_b, hipHostRegisterDefault); double r1 = now();
    double t0 = now(); (void)hipMemcpyAsync(din, pin, in_b, hipMemcpyHostToDevice, s); (void)hipStreamSynchronize(s); double t1 = now();
    (void)hipMemcpyAsync(pout, dout, out_b, hipMemcpyDeviceToHost, s); (void)hipStreamSynchronize(s); double t2 = now();
    double u0 = now(); (void)hipHostUnregister(pin); (void)hipHostUnregister(pout); double u1 = now();
    printf("registered (%d,%d) register %.1f ms  H2D %.1f ms (%.1f GB/s)  D2H %.1f ms (%.1f GB/s)  unregister %.1f ms\n", (int)e1, (int)e2, (r1 - r0) * 1e3,
           (t1 - t0) * 1e3, in_b / (t1 - t0) / 1e9, (t2 - t1) * 1e3, out_b / (t2 - t1) / 1e9, (u1 - u0) * 1e3);
  }
  return 0;
}
