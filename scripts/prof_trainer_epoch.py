"""cProfile of the main thread during one epoch of the module path (bench.trainer_epoch_section's set-up)."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                   # noqa: E402
from idiaptts_amd.src.neural_networks.pytorch import ModularModelHandlerPyTorch as M    # noqa: E402

H = M.ModularModelHandlerPyTorch
calls = {"n": 0}
orig = H.train


def traced(self, *a, **kw):
    calls["n"] += 1
    if calls["n"] == 2:
        pr = cProfile.Profile()
        pr.enable()
        r = orig(self, *a, **kw)
        torch.cuda.synchronize()
        pr.disable()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
        return r
    return orig(self, *a, **kw)


H.train = traced
os.environ["ITTS_TRAINER_EPOCH_ONLY"] = "module_path"
print(bench.trainer_epoch_section(torch.device("cuda", 0)))
