#!/bin/bash
# how much of the pulse kernels is their overlap-add's atomics?  (SYN_WAVE_DIAG=1: plain conditional stores instead)
cd "$GRAFT_REPO_ROOT" || exit 1
L=idiaptts_amd/_lib
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=on -munsafe-fp-atomics"
for def in "-DSYN_WAVE_DIAG=1" ""; do
  /opt/rocm/bin/hipcc $FLAGS $def -c idiaptts_amd/csrc/synth.hip -o $L/synth.o 2>/dev/null || exit 2
  /opt/rocm/bin/hipcc -shared -fPIC -pthread --offload-arch=gfx950 -o $L/libidiaptts_amd.so $L/*.o || exit 3
  echo "== ${def:-atomics}"
  bash scripts/syn_timeline.sh 2>&1 | grep "pulse_wave"
done
