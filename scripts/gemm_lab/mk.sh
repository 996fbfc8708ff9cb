#!/bin/bash
# mk.sh <output name> [extra hipcc flags]: builds one lab variant next to this script
cd "$(dirname "$0")"
out=$1; shift
hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=on -munsafe-fp-atomics lab.hip -o $out \
  -L../../idiaptts_amd/_lib -lidiaptts_amd -Wl,-rpath,'$ORIGIN/../../idiaptts_amd/_lib' "$@" 2>&1 | grep -A6 "error" 
exit 0
