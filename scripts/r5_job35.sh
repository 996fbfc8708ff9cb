#!/bin/bash
R=$GRAFT_REPO_ROOT; cd "$R" || exit 1
O=$R/gpurun_out/r5ad; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_world.py tests/test_gpu_properties.py tests/test_gpu_fuzz.py tests/test_gpu_loader.py tests/test_gpu_dropin.py tests/test_gpu_trainer.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest.txt
bash scripts/syn_timeline.sh 2>&1 | tail -16
