// Developer tool: triangulates the out-of-line homogeneous addition step mismatch.  Runs the same step as
//   0 inline   1 out-of-line by reference (the failing shape)   2 out-of-line by value   3 out-of-line by reference with
//   the point COPIED OUT through a local before the in/out argument is written   4 inline + intermediates   5 out-of-line +
//   intermediates
// on one wave and dumps points and intermediates as raw 28-bit limbs (16 dwords per lane and value) for tools/kbench/dbg_hom2.py.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "pairing.hip.h"
using namespace bls;

struct Dump {
  uint32_t* base; int slot;
  __device__ void put(int k, const Fp2R& a) const {
    uint32_t* d = base + ((size_t)(slot * 16 + k) * 64 + threadIdx.x) * 16;
    for (int i = 0; i < NL; ++i) d[i] = a.v.l[i];
  }
};
struct NoDump { __device__ void put(int, const Fp2R&) const {} };

template <class D>
BLS_HD LineCoeffs step_body(G2Jac& r_io, const G2Aff& q, const D& dump) {
  const G2Jac r = r_io;
  const auto theta = fit<1, 2>(sub(r.y, mul(q.y, r.z)));
  const auto lambda = fit<1, 2>(sub(r.x, mul(q.x, r.z)));
  const Fp2R c = sqr(theta);
  const Fp2R d = sqr(lambda);
  const Fp2R e = mul(lambda, d);
  const Fp2R f = mul(r.z, c);
  const Fp2R g = mul(r.x, d);
  const auto h = fit<1, 2>(sub(add(e, f), dbl(g)));
  const auto j = fit<1, 2>(sub(mul(theta, q.x), mul(lambda, q.y)));
  dump.put(0, Fp2R(theta)); dump.put(1, Fp2R(lambda)); dump.put(2, c); dump.put(3, d); dump.put(4, e); dump.put(5, f);
  dump.put(6, g); dump.put(7, Fp2R(h)); dump.put(8, Fp2R(j));
  LineCoeffs l;
  l.c = j;
  l.b = fit<1, 2>(neg(theta));
  l.a = lambda;
  G2Jac o;
  o.x = mul(lambda, h);
  o.y = fit<1, 2>(sub(mul(theta, sub(g, h)), mul(e, r.y)));
  o.z = mul(r.z, e);
  dump.put(9, o.x); dump.put(10, o.y); dump.put(11, o.z);
  r_io = o;
  return l;
}

BLS_HD_NOINLINE LineCoeffs step_call_ref(G2Jac& r, const G2Aff& q) { return addition_step_hom(r, q); }
struct StepOut { G2Jac r; LineCoeffs l; };
BLS_HD_NOINLINE StepOut step_call_val(G2Jac r, G2Aff q) { StepOut o; o.l = addition_step_hom(r, q); o.r = r; return o; }
BLS_HD_NOINLINE LineCoeffs step_call_dump(G2Jac& r, const G2Aff& q, Dump d) { return step_body(r, q, d); }
BLS_HD_NOINLINE LineCoeffs step_call_nodump(G2Jac& r, const G2Aff& q) { return step_body(r, q, NoDump()); }

__device__ void put_pt(uint32_t* base, int slot, const G2Jac& r) {
  Dump d{base, slot}; d.put(12, r.x); d.put(13, r.y); d.put(14, r.z);
}
__device__ void put_line(uint32_t* base, int slot, const LineCoeffs& l) {
  Dump d{base, slot}; d.put(13, l.a); d.put(14, l.b); d.put(15, l.c);
}

__global__ void k(const uint64_t* g2, uint32_t* out) {
  const size_t i = threadIdx.x / 2;
  bool inf;
  G2Aff q = g2_aff_load(g2 + 24 * i, 0, inf);
  G2Jac r0; r0.x = q.x; r0.y = q.y; r0.z = Fp2R(fp2_one());
  (void)doubling_step_hom(r0);
  { Dump d{out, 7}; d.put(0, q.x); d.put(1, q.y); d.put(12, r0.x); d.put(13, r0.y); d.put(14, r0.z); }
  { G2Jac r = r0; (void)addition_step_hom(r, q); put_pt(out, 0, r); }
  { G2Jac r = r0; (void)step_call_ref(r, q); put_pt(out, 1, r); }
  { StepOut o = step_call_val(r0, q); put_pt(out, 2, o.r); }
  { G2Jac r = r0; (void)step_call_nodump(r, q); put_pt(out, 3, r); }
  { G2Jac r = r0; (void)step_body(r, q, Dump{out, 4}); put_pt(out, 4, r); }
  { G2Jac r = r0; (void)step_call_dump(r, q, Dump{out, 5}); put_pt(out, 5, r); }
}

int main() {
  std::vector<uint64_t> h(24 * 32);
  uint64_t s = 12345; for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = s; }
  for (size_t i = 5; i < h.size(); i += 6) h[i] &= 0x0fffffffffffffffull;
  uint64_t* d; uint32_t* o; const size_t words = 8 * 16 * 64 * 16;
  hipMalloc(&d, h.size() * 8); hipMalloc(&o, words * 4); hipMemset(o, 0, words * 4);
  hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  k<<<1, 64>>>(d, o);
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) { printf("ERROR %s\n", hipGetErrorString(e)); return 1; }
  std::vector<uint32_t> r(words); hipMemcpy(r.data(), o, words * 4, hipMemcpyDeviceToHost);
  // text dump: slot value lane limbs...   (lanes 0 and 1 = first element only, plus lanes 2,3)
  for (int slot = 0; slot < 8; ++slot) for (int v = 0; v < 16; ++v) for (int lane = 0; lane < 4; ++lane) {
    printf("%d %d %d", slot, v, lane);
    for (int i = 0; i < 14; ++i) printf(" %x", r[((size_t)(slot * 16 + v) * 64 + lane) * 16 + i]);
    printf("\n");
  }
  return 0;
}
