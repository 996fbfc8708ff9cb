#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5h; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_world.py -m gpu -x -q > $O/pytest.txt 2>&1
echo "pytest rc $?" >> $O/pytest.txt
for g in 0 1; do
  if [ $g = 1 ]; then export ITTS_D4C_GENERIC=1; else unset ITTS_D4C_GENERIC; fi
  SERIAL=1 bash scripts/analysis_prof.sh r5h_g$g 256 16000 > $O/analysis_g$g.txt 2>&1
  SERIAL=1 bash scripts/analysis_prof.sh r5h_g$g 64 48000 > $O/analysis48_g$g.txt 2>&1
done
tail -3 $O/pytest.txt; grep -E "d4c_kernel|total kernel" $O/analysis_g0.txt $O/analysis_g1.txt $O/analysis48_g0.txt $O/analysis48_g1.txt
