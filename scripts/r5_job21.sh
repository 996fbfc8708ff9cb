#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1200 python -m pytest tests/test_gpu_loader.py tests/test_gpu_trainer.py tests/test_gpu_dropin.py tests/test_gpu_model.py tests/test_gpu_dp.py -m gpu -x -q 2>&1 | tail -3
bash scripts/r5_job20.sh 2>&1 | grep -A6 "module_path\|resident" | grep "epoch_s_all" -A3 | head -20
