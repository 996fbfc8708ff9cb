#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5k; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_dropin.py tests/test_gpu_model.py tests/test_gpu_trainer.py tests/test_hostio.py -m gpu -x -q > $O/pytest.txt 2>&1
echo "pytest rc $?" >> $O/pytest.txt
for cfg in "128 64" "512 64" "512 128"; do
  echo "== $cfg" >> $O/gen_data.txt
  ITTS_GEN_DATA_TRACE=1 timeout 300 python3 scripts/prof_gen_data.py $cfg 2>&1 | tail -12 >> $O/gen_data.txt
done
tail -3 $O/pytest.txt; grep -E "==|writers|rtf" $O/gen_data.txt
