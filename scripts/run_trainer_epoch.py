"""bench.trainer_epoch_section alone (rows chosen by ITTS_TRAINER_EPOCH_ONLY), a short table of its rows."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                   # noqa: E402

r = bench.trainer_epoch_section(torch.device("cuda", 0), int(sys.argv[1]) if len(sys.argv) > 1 else 1024)["trainer_epoch"]
keep = ("epoch_s_all", "valid_frames_per_s", "ms_per_step", "train_call_s", "device_cache")
print(json.dumps({k: (v if not isinstance(v, dict) else {kk: v[kk] for kk in keep if kk in v}) for k, v in r.items()},
                 indent=1))
