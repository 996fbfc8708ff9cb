#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5o; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_world.py -m gpu -x -q -k "d4c or world or analysis or cmp or 48" 2>&1 | tail -3 > $O/pytest.txt
for g in 0 1; do
  if [ $g = 1 ]; then export ITTS_D4C_NO_PLAN=1; else unset ITTS_D4C_NO_PLAN; fi
  SERIAL=1 bash scripts/analysis_prof.sh r5o_p$g 256 16000 > $O/analysis_p$g.txt 2>&1
done
cat $O/pytest.txt; grep -E "d4c_kernel|total kernel" gpurun_out/r5o_p0_analysis_kstats_16000.txt gpurun_out/r5o_p1_analysis_kstats_16000.txt
