#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5z; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_mlpg.py tests/test_gpu_fuzz.py tests/test_gpu_model.py tests/test_gpu_dropin.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest2.txt
for v in "1,1,1" "1,0.1,0.05" "1,0.03,0.03" "1,0.01,0.01" ""; do
  for a in "100 256 f64" "30 4096 f64" "30 4096 f32"; do
    echo -n "var $v  " | tee -a $O/unified.txt
    MLPG_TIME_VAR=$v timeout 300 python3 scripts/mlpg_time.py $a 2>&1 | tail -1 | sed -E 's/ p90.*\|/ |/; s/, host.*//' | tee -a $O/unified.txt
  done
done
