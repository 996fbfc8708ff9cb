#!/bin/bash
# A/B: the head's rows for the way back staged in the ring (<= 56 rows) against always read from the table
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5z; mkdir -p $O
L=idiaptts_amd/_lib
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=on -munsafe-fp-atomics"
for def in "" "-DMLPG_RING_NO_STAGE" "" "-DMLPG_RING_NO_STAGE"; do
  /opt/rocm/bin/hipcc $FLAGS $def -c idiaptts_amd/csrc/mlpg.hip -o $L/mlpg.o || exit 2
  /opt/rocm/bin/hipcc -shared -fPIC -pthread --offload-arch=gfx950 -o $L/libidiaptts_amd.so $L/*.o || exit 3
  echo "== ${def:-staged}" | tee -a $O/stage.txt
  for v in "1,1,1" "1,0.1,0.05"; do
    for a in "100 256 f64" "30 4096 f64" "30 4096 f32"; do
      echo -n "var $v  " | tee -a $O/stage.txt
      MLPG_TIME_VAR=$v timeout 300 python3 scripts/mlpg_time.py $a 2>&1 | tail -1 | sed -E 's/ p90.*\|/ |/; s/, host.*//' | tee -a $O/stage.txt
    done
  done
done
timeout 600 python -m pytest tests/test_gpu_mlpg.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -2
