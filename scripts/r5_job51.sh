#!/bin/bash
# cheaptrick_wave_kernel: frames from a counter, CTW_DEAL neighbours at a time (0: by stride)
R=$GRAFT_REPO_ROOT; cd "$R" || exit 1
O=$R/gpurun_out/r5av; mkdir -p $O
L=idiaptts_amd/_lib
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=on -munsafe-fp-atomics"
cd /tmp; export TMPDIR=/tmp
for v in 0 1 2 4 8; do
  ( cd $R && /opt/rocm/bin/hipcc $FLAGS -DCTW_DEAL=$v -c idiaptts_amd/csrc/world_frame.hip -o $L/world_frame.o 2>/dev/null && /opt/rocm/bin/hipcc -shared -fPIC -pthread --offload-arch=gfx950 -o $L/libidiaptts_amd.so $L/*.o ) || exit 3
  for fs in 16000 48000; do
    rm -rf /tmp/ak; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ak -- python3 $R/scripts/traffic_driver.py analysis 4 $fs > /tmp/ak.log 2>&1
    echo "== CTW_DEAL=$v fs=$fs: $(python3 $R/scripts/kstats.py /tmp/ak 30 2>/dev/null | grep cheaptrick_wave)" | tee -a $O/ctw_deal_ab.txt
  done
done
cd $R
timeout 900 python -m pytest tests/test_gpu_world.py tests/test_gpu_properties.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest.txt
