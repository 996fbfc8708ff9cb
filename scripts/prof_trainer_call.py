"""cProfile of a whole `trainer.train` call of bench.trainer_epoch_section (ITTS_TRAINER_EPOCH_ONLY picks the row)."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                   # noqa: E402
from idiaptts_amd.src.model_trainers import ModularTrainer as MT    # noqa: E402

orig = MT.ModularTrainer.train


def traced(self, *a, **kw):
    pr = cProfile.Profile()
    pr.enable()
    r = orig(self, *a, **kw)
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
    return r


MT.ModularTrainer.train = traced
os.environ.setdefault("ITTS_TRAINER_EPOCH_ONLY", "resident_dataset")
r = bench.trainer_epoch_section(torch.device("cuda", 0))["trainer_epoch"]
print({k: v for k, v in r.items() if not isinstance(v, dict)}, {k: v.get("train_call_s") for k, v in r.items() if isinstance(v, dict)})
