#!/bin/bash
# Builds kbench variants in parallel: each line "name flags...".
I=../../plonky2-bls12-381-pairing_amd/csrc
build() { name=$1; shift; ( s=$(date +%s); hipcc -O3 -std=c++17 --offload-arch=gfx950 -I$I -DKB_NAME="\"$name\"" "$@" kbench.hip -o kb_$name 2>&1 | grep -E "error" | head -3; echo "$name built in $(( $(date +%s) - s ))s" ) & }
while read -r line; do [ -n "$line" ] && build $line; done
wait
