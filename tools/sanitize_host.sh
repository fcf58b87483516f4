#!/bin/bash
# Host-side sanitizer pass (GPU sanitizers are not available on this pool): builds the C oracle and the host
# build of the device headers with -fsanitize=address,undefined and runs the CPU parity tests against them.
set -e
cd "$(dirname "$0")/.."
INC=plonky2-bls12-381-pairing_amd/csrc
mkdir -p /tmp/bls_san
gcc -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -fPIC -shared oracle/bls381_oracle.c -o /tmp/bls_san/libbls381_oracle.so -lpthread
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -fPIC -shared -I $INC tests/hostsim/hostsim.cpp -o /tmp/bls_san/libhostsim.so
cp oracle/libbls381_oracle.so /tmp/bls_san/oracle.bak 2>/dev/null || true
cp tests/hostsim/libhostsim.so /tmp/bls_san/hostsim.bak 2>/dev/null || true
cp /tmp/bls_san/libbls381_oracle.so oracle/libbls381_oracle.so
cp /tmp/bls_san/libhostsim.so tests/hostsim/libhostsim.so
touch oracle/libbls381_oracle.so tests/hostsim/libhostsim.so
trap 'cp /tmp/bls_san/oracle.bak oracle/libbls381_oracle.so 2>/dev/null; cp /tmp/bls_san/hostsim.bak tests/hostsim/libhostsim.so 2>/dev/null; touch oracle/libbls381_oracle.so tests/hostsim/libhostsim.so' EXIT
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0 \
  python -m pytest tests/test_oracle_c.py tests/test_hostsim.py tests/test_witness_hostsim.py -x -q
