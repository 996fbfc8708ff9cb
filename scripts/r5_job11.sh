#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5l; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_world.py -m gpu -x -q > $O/pytest.txt 2>&1
echo "pytest rc $?" >> $O/pytest.txt
SERIAL=1 bash scripts/analysis_prof.sh r5l 256 16000 > $O/analysis.txt 2>&1
SERIAL=1 bash scripts/analysis_prof.sh r5l 64 48000 > $O/analysis48.txt 2>&1
tail -3 $O/pytest.txt; grep -E "stonemask|cheaptrick_wave|total kernel" gpurun_out/r5l_analysis_kstats_16000.txt gpurun_out/r5l_analysis_kstats_48000.txt
