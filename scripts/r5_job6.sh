#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_world.py tests/test_gpu_dropin.py tests/test_gpu_mgc.py -m gpu -x -q > $O/pytest.txt 2>&1
echo "pytest rc $?" >> $O/pytest.txt
for w in 8; do
  ITTS_F3_WAVES=$w SERIAL=1 bash scripts/analysis_prof.sh r5g_w$w 256 16000 > $O/analysis_w$w.txt 2>&1
done
tail -3 $O/pytest.txt; grep -E "fused|total kernel" $O/analysis_w8.txt
