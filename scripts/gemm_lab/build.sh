#!/bin/bash
# builds the GEMM lab binary next to this script (links the in-tree library for the A/B baseline)
set -e
cd "$(dirname "$0")"
LIB=../../idiaptts_amd/_lib
hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=on -munsafe-fp-atomics lab.hip -o lab \
  -L$LIB -lidiaptts_amd -Wl,-rpath,'$ORIGIN/../../idiaptts_amd/_lib' "$@"
