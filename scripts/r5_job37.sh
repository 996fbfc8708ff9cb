#!/bin/bash
R=$GRAFT_REPO_ROOT; cd "$R" || exit 1
O=$R/gpurun_out/r5af; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_world.py tests/test_gpu_properties.py tests/test_gpu_loader.py tests/test_gpu_dropin.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest.txt
cd /tmp; export TMPDIR=/tmp
for fs in 48000 16000; do
  rm -rf /tmp/sk; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sk -- python3 $R/scripts/traffic_driver.py synthesis 4 $fs > /tmp/sk.log 2>&1
  python3 $R/scripts/kstats.py /tmp/sk 2>/dev/null | grep -i "syn_\|decode_ap\|gemm_f64_kernel<true, false, true>\|total" | head -14 | tee $O/synthesis_${fs}_kstats.txt
done
