#!/bin/bash
# the pulses of syn_pulse_wave_kernel dealt behind a counter, SYN_WAVE_DEAL at a time (0: by stride, as before)
R=$GRAFT_REPO_ROOT; cd "$R" || exit 1
O=$R/gpurun_out/r5ar; mkdir -p $O
L=idiaptts_amd/_lib
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=on -munsafe-fp-atomics"
for deal in "4 -DSYN_WAVE_DEAL_VOI=1" "6 -DSYN_WAVE_DEAL_VOI=3" "3 -DSYN_WAVE_DEAL_VOI=2" "4 -DSYN_WAVE_DEAL_VOI=2"; do
  /opt/rocm/bin/hipcc $FLAGS -DSYN_WAVE_DEAL_UNV=$deal -c idiaptts_amd/csrc/synth.hip -o $L/synth.o 2>/dev/null || exit 2
  /opt/rocm/bin/hipcc -shared -fPIC -pthread --offload-arch=gfx950 -o $L/libidiaptts_amd.so $L/*.o || exit 3
  echo "== UNV=$deal" | tee -a $O/deal_ab.txt
  bash scripts/syn_timeline.sh 2>&1 | grep -E "pulse_wave|synthesis:" | tee -a $O/deal_ab.txt
done
timeout 900 python -m pytest tests/test_gpu_world.py tests/test_gpu_loader.py tests/test_gpu_dropin.py tests/test_gpu_properties.py tests/test_gpu_trainer.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest.txt
