#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5q; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_mlpg.py tests/test_gpu_fuzz.py tests/test_gpu_dropin.py tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -4 > $O/pytest.txt
cat $O/pytest.txt
for n in 256 4096; do
python3 scripts/mlpg_time.py 40 $n
python3 scripts/mlpg_time.py 40 $n f32
done
