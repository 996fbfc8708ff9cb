#!/bin/bash
R=$GRAFT_REPO_ROOT; cd "$R" || exit 1
O=$R/gpurun_out/r5ai; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_world.py tests/test_gpu_properties.py tests/test_gpu_dropin.py tests/test_gpu_fuzz.py tests/test_gpu_select.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest.txt
cd /tmp; export TMPDIR=/tmp
for fs in 16000 48000; do
  rm -rf /tmp/ak; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ak -- python3 $R/scripts/traffic_driver.py analysis 4 $fs > /tmp/ak.log 2>&1
  python3 $R/scripts/kstats.py /tmp/ak 2>/dev/null | head -9 | tee $O/analysis_${fs}_kstats.txt
done
