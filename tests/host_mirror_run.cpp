// GPU test driver for the C++ mirror of the reference interface (bls381_pairing.hpp): reads
//   G1Projective (144 B) | G2Projective (288 B) | G1Affine x,y (96 B) | G2Affine x,y (192 B)
// from argv[1], writes optimized_miller_loop(P, Q) and multi_miller_loop([(Pa, Qa)]).final_exponentiation()
// (2 x 576 B) to argv[2].  tests/test_gpu_parity.py compares the bytes with the oracle.
#include <cstdio>
#include <cstring>
#include <vector>
#include "../plonky2-bls12-381-pairing_amd/host/bls381_pairing.hpp"

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  std::vector<unsigned char> in(144 + 288 + 96 + 192);
  FILE* f = fopen(argv[1], "rb");
  if (!f || fread(in.data(), 1, in.size(), f) != in.size()) return 3;
  fclose(f);
  bls381::G1Projective P; bls381::G2Projective Q; bls381::G1Affine pa; bls381::G2Affine qa;
  std::memcpy(&P, in.data(), 144);
  std::memcpy(&Q, in.data() + 144, 288);
  std::memcpy(&pa.x, in.data() + 432, 48); std::memcpy(&pa.y, in.data() + 480, 48);
  std::memcpy(&qa.x, in.data() + 528, 96); std::memcpy(&qa.y, in.data() + 624, 96);
  try {
    bls381::Backend be(0);
    const bls381::Fq12 a = bls381::optimized_miller_loop(be, P, Q);                       // opt.rs:81
    const bls381::MillerLoopResult m = bls381::multi_miller_loop(be, {{&pa, &qa}});       // miller_loop_native.rs:154
    const bls381::Fq12 e = m.final_exponentiation(be);
    const bls381::MillerLoopResult one = bls381::multi_miller_loop(be, {});               // literal behaviour: ONE
    FILE* o = fopen(argv[2], "wb");
    fwrite(&a, 1, 576, o); fwrite(&e, 1, 576, o); fwrite(&one.value, 1, 576, o);
    fclose(o);
  } catch (const bls381::PairingError& ex) {
    fprintf(stderr, "PairingError: %s\n", ex.what());
    return 1;
  }
  return 0;
}
