#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5c; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_world.py -m gpu -x -q -k "mcep or newton or cheaptrick or features" > $O/pytest.txt 2>&1
echo "pytest rc $?" >> $O/pytest.txt
for m in 0 1; do
  ITTS_MCEP_FUSED=$m SERIAL=1 bash scripts/analysis_prof.sh r5c_fused$m 256 16000 > $O/analysis_fused$m.txt 2>&1
done
timeout 120 ./scripts/xcd_lab/xcd_split 5 > $O/xcd_split.txt 2>&1
tail -3 $O/pytest.txt; grep -E "fused|gemm_f64|total kernel" $O/analysis_fused0.txt $O/analysis_fused1.txt; cat $O/xcd_split.txt
