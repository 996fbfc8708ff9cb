#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for n in 16 64 256 1024 4096; do
python3 scripts/mlpg_time.py 30 $n
ITTS_MLPG_STREAM=1 python3 scripts/mlpg_time.py 30 $n
done
