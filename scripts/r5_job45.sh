#!/bin/bash
# the overlap-add of a workgroup's four pulses summed in LDS before the atomics (SYN_WAVE_COMBINE=1) against each wave its own (=0)
R=$GRAFT_REPO_ROOT; cd "$R" || exit 1
O=$R/gpurun_out/r5ao; mkdir -p $O
L=idiaptts_amd/_lib
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=on -munsafe-fp-atomics"
for def in "-DSYN_WAVE_COMBINE=0" "-DSYN_WAVE_COMBINE=1"; do
  /opt/rocm/bin/hipcc $FLAGS $def -c idiaptts_amd/csrc/synth.hip -o $L/synth.o 2>/dev/null || exit 2
  /opt/rocm/bin/hipcc -shared -fPIC -pthread --offload-arch=gfx950 -o $L/libidiaptts_amd.so $L/*.o || exit 3
  echo "== $def" | tee -a $O/combine_ab.txt
  bash scripts/syn_timeline.sh 2>&1 | grep -E "pulse_wave|synthesis:" | tee -a $O/combine_ab.txt
done
timeout 900 python -m pytest tests/test_gpu_world.py tests/test_gpu_loader.py tests/test_gpu_dropin.py tests/test_gpu_properties.py -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest.txt
