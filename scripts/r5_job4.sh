#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5d; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_world.py -m gpu -x -q -k "mcep or newton or cheaptrick or features" > $O/pytest.txt 2>&1
echo "pytest rc $?" >> $O/pytest.txt
for m in 2 3; do
  ITTS_MCEP_FUSED=$m SERIAL=1 bash scripts/analysis_prof.sh r5d_fused$m 256 16000 > $O/analysis_fused$m.txt 2>&1
done
ITTS_MCEP_FUSED=3 SERIAL=1 bash scripts/analysis_prof.sh r5d_fused3 64 48000 > $O/analysis48_fused3.txt 2>&1
timeout 300 python scripts/load_step_probe.py > $O/load_step_probe.txt 2>&1
tail -5 $O/pytest.txt; grep -E "fused|gemm_f64|total kernel" $O/analysis_fused2.txt $O/analysis_fused3.txt $O/analysis48_fused3.txt; cat $O/load_step_probe.txt
