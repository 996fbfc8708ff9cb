#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1200 python -m pytest tests/test_gpu_world.py tests/test_gpu_select.py -m gpu -x -q 2>&1 | tail -3
bash scripts/analysis_prof.sh r5t 64 48000 2>&1 | grep -i "d4c_kernel\|total kernel"
bash scripts/analysis_prof.sh r5t 256 16000 2>&1 | grep -i "d4c_kernel\|total kernel"
