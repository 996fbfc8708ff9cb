#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5ab; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_mlpg.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -2 | tee $O/pytest.txt
for a in "100 256 f64" "30 1024 f64" "30 4096 f64" "30 4096 f32"; do
  timeout 300 python3 scripts/mlpg_time.py $a 2>&1 | tail -1 | tee -a $O/time.txt
done
