#!/bin/bash
# A/B of the cache policy of mlpg_ring_kernel's streams (MLPG_RING_NT bits), every variant on the same box,
# the same variances (seeded) in every run
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5w; mkdir -p $O
L=idiaptts_amd/_lib
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=on -munsafe-fp-atomics"
run() {
  echo "== MLPG_RING_NT=$1" | tee -a $O/nt.txt
  for a in "100 256 f64" "30 4096 f64" "100 256 f32" "30 4096 f32"; do
    timeout 300 python3 scripts/mlpg_time.py $a 2>&1 | tail -1 | tee -a $O/nt.txt
  done
}
for nt in ${NTS:-0 1 3 7 15 5 0}; do
  /opt/rocm/bin/hipcc $FLAGS -DMLPG_RING_NT=$nt -c idiaptts_amd/csrc/mlpg.hip -o $L/mlpg.o || exit 2
  /opt/rocm/bin/hipcc -shared -fPIC -pthread --offload-arch=gfx950 -o $L/libidiaptts_amd.so $L/*.o || exit 3
  run $nt
done
