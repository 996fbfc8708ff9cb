#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5q; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_mlpg.py tests/test_gpu_fuzz.py -m gpu -x -q -k mlpg 2>&1 | tail -4 > $O/pytest.txt
cat $O/pytest.txt
ITTS_MLPG_WIDE=1 timeout 900 python -m pytest tests/test_gpu_mlpg.py tests/test_gpu_fuzz.py -m gpu -x -q -k "mlpg and not wide" 2>&1 | tail -4
