#!/bin/bash
# round 5, job 2: fused Newton products (parity + time), trainer_epoch row, the small ADVICE items
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_world.py tests/test_gpu_nn.py tests/test_gpu_dp.py tests/test_gpu_dropin.py tests/test_abi.py -m gpu -x -q > $O/pytest.txt 2>&1
echo "pytest rc $?" >> $O/pytest.txt
for m in 0 1; do
  ITTS_MCEP_FUSED=$m SERIAL=1 bash scripts/analysis_prof.sh r5b_fused$m 256 16000 > $O/analysis_fused$m.txt 2>&1
done
ITTS_MCEP_FUSED=1 SERIAL=1 bash scripts/analysis_prof.sh r5b_fused1 64 48000 > $O/analysis48_fused1.txt 2>&1
timeout 600 python - > $O/trainer_epoch.json 2> $O/trainer_epoch.err <<'PY'
import json, torch, bench
print(json.dumps(bench.trainer_epoch_section(torch.device("cuda", 0), 256)))
PY
tail -3 $O/pytest.txt; tail -12 $O/analysis_fused0.txt; tail -12 $O/analysis_fused1.txt; cat $O/trainer_epoch.json; tail -3 $O/trainer_epoch.err
