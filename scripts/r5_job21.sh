#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_gpu_loader.py tests/test_gpu_trainer.py tests/test_gpu_dropin.py -m gpu -x -q 2>&1 | tail -4
bash scripts/r5_job20.sh 2>&1 | grep -A6 "module_path\|resident" | head -40
