#!/bin/bash
R=$GRAFT_REPO_ROOT; cd "$R" || exit 1
timeout 1500 python -m pytest tests/test_gpu_world.py tests/test_gpu_mgc.py tests/test_gpu_properties.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -2
O=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/sp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp -- python3 $R/scripts/traffic_driver.py synthesis 4 16000 256 > /dev/null 2>&1
python3 $R/scripts/kstats.py /tmp/sp 20 | grep "gemm_f64\|total"
