#!/bin/bash
# syn_pulse_kernel (48 kHz): a stretch of the pulse list per XCD (SYN_PULSE_XCD=1) against pulse = blockIdx (=0)
R=$GRAFT_REPO_ROOT; cd "$R" || exit 1
O=$R/gpurun_out/r5au; mkdir -p $O
L=idiaptts_amd/_lib
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=on -munsafe-fp-atomics"
cd /tmp; export TMPDIR=/tmp
for v in 0 1; do
  ( cd $R && /opt/rocm/bin/hipcc $FLAGS -DSYN_PULSE_XCD=$v -c idiaptts_amd/csrc/synth.hip -o $L/synth.o 2>/dev/null && /opt/rocm/bin/hipcc -shared -fPIC -pthread --offload-arch=gfx950 -o $L/libidiaptts_amd.so $L/*.o ) || exit 3
  echo "== SYN_PULSE_XCD=$v" | tee -a $O/xcd_ab.txt
  rm -rf /tmp/sk; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sk -- python3 $R/scripts/traffic_driver.py synthesis 4 48000 > /tmp/sk.log 2>&1
  python3 $R/scripts/kstats.py /tmp/sk 2>/dev/null | head -6 | tee -a $O/xcd_ab.txt
done
cd $R
timeout 900 python -m pytest tests/test_gpu_world.py tests/test_gpu_loader.py tests/test_gpu_dropin.py tests/test_gpu_properties.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest.txt
