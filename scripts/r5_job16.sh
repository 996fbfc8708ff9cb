#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5q; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_mlpg.py tests/test_gpu_fuzz.py tests/test_gpu_dropin.py tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -6 > $O/pytest.txt
cat $O/pytest.txt
bash scripts/mlpg_timeline.sh | tail -3
for n in 16 64 256 1024 4096; do
python3 scripts/mlpg_time.py 30 $n
ITTS_MLPG_STREAM=1 python3 scripts/mlpg_time.py 30 $n
done
