#!/bin/bash
# One kbench variant per directory (device assembly kept for tools/kbench/asm_stats.py):
#   tools/kbench/build_variant.sh <name> [-D...]   ->  tools/kbench/out/<name>/kb  (+ *.s)
# Flags match the library build (plonky2-bls12-381-pairing_amd/Makefile).
set -e
cd "$(dirname "$0")"
name=$1; shift
src=${KB_SRC:-kbench.hip}
mkdir -p out/$name
hipcc -O3 -std=c++17 --offload-arch=gfx950 -DBLS_LANES=2 -mllvm -amdgpu-dpp-combine=false \
  -I../../plonky2-bls12-381-pairing_amd/csrc -DKB_NAME="\"$name\"" "$@" -save-temps=obj $src -o out/$name/kb > out/$name/build.log 2>&1 \
  || { tail -20 out/$name/build.log; exit 1; }
rm -f out/$name/*.bc out/$name/*.hipi out/$name/*.out out/$name/*.hipfb out/$name/*host*.s out/$name/*.resolution.txt out/$name/*.o
python3 ../check_device_asm.py out/$name/*gfx950.s
echo "built $name"
