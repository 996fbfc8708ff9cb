#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5j; mkdir -p $O
for cfg in "128 64" "256 64" "512 64" "512 32" "512 128"; do
  echo "== $cfg" >> $O/gen_data.txt
  ITTS_GEN_DATA_TRACE=1 timeout 300 python3 scripts/prof_gen_data.py $cfg 2>&1 | tail -14 >> $O/gen_data.txt
done
cat $O/gen_data.txt
