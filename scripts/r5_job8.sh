#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5i; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_world.py tests/test_gpu_dropin.py tests/test_gpu_mgc.py -m gpu -x -q > $O/pytest.txt 2>&1
echo "pytest rc $?" >> $O/pytest.txt
SERIAL=1 bash scripts/analysis_prof.sh r5i 256 16000 > $O/analysis.txt 2>&1
SERIAL=1 bash scripts/analysis_prof.sh r5i 64 48000 > $O/analysis48.txt 2>&1
tail -3 $O/pytest.txt; head -20 gpurun_out/r5i_analysis_kstats_16000.txt
