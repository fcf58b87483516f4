// Compile-only check that the C++ mirror of the reference interface is self-consistent (tests/test_abi.py).
#include "../plonky2-bls12-381-pairing_amd/host/bls381_pairing.hpp"
int main() {
  try { bls381::Backend be(0); (void)bls381::MillerLoopResult::one(be); } catch (const bls381::PairingError&) { return 0; }
  return 0;
}
