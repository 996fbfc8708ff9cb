#!/bin/bash
# MLPG one-pass solve against the length of the factor's head (set through the variances)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5w; mkdir -p $O
for v in "1,1,1" "1,0.3,0.3" "1,0.1,0.05" "1,0.07,0.05" "1,0.05,0.05" "1,0.03,0.03" "1,0.01,0.01" "1,0.003,0.003"; do
  for a in "100 256 f64" "30 4096 f64" "30 4096 f32"; do
    echo -n "var $v  " | tee -a $O/head.txt
    MLPG_TIME_VAR=$v timeout 300 python3 scripts/mlpg_time.py $a 2>&1 | tail -1 | sed -E 's/ p90.*//' | tee -a $O/head.txt
  done
done
