#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 600 python3 - <<'PY'
import json, torch, bench
dev = torch.device("cuda", 0)
r = bench.trainer_epoch_section(dev)
r = r.get("trainer_epoch", r)
print(json.dumps({k: (v if not isinstance(v, dict) else {kk: v[kk] for kk in ("epoch_s_all", "valid_frames_per_s", "train_call_s", "dataloader_workers") if kk in v}) for k, v in r.items()}, indent=1))
PY
