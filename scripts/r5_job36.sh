#!/bin/bash
R=$GRAFT_REPO_ROOT; cd "$R" || exit 1
python3 -c "
import torch
print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else 'n/a')
for p in (-1,0,1,2):
    s=torch.cuda.Stream(priority=p); print(p, s.priority)
"
for pr in 1 0 -1; do
  echo "== side priority $pr"
  ITTS_SIDE_PRIORITY=$pr bash scripts/syn_timeline.sh 2>&1 | grep "gemm_f64\|phase_scan\|decode_ap\|pulse_wave\|randn"
done
