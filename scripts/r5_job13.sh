#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5n; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_dropin.py tests/test_hostio.py -m gpu -x -q 2>&1 | tail -3 > $O/pytest.txt
for cfg in "512 64" "512 128"; do
  echo "== $cfg" >> $O/gen_data.txt
  ITTS_GEN_DATA_TRACE=1 timeout 300 python3 scripts/prof_gen_data.py $cfg --dir=/dev/shm 2>&1 | tail -14 >> $O/gen_data.txt
done
cat $O/pytest.txt; grep -E "==|writers|rtf|batch [0-9]+:" $O/gen_data.txt | tail -40
