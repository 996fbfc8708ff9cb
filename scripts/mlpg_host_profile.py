"""Where the host side of ops.mlpg_generation goes (256 utterances): cProfile over queued calls."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from idiaptts_amd import ops, world                             # noqa: E402
from idiaptts_amd.bench_support import utterance_lengths       # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
off = world.offsets(utterance_lengths(256, seed=5).tolist())
feat = torch.randn(off[-1], 186, dtype=torch.float64, device=dev)
var = torch.rand(186, dtype=torch.float64, device=dev) * 0.99 + 0.01
for _ in range(20):
    ops.mlpg_generation(feat, var, 62, off)
torch.cuda.synchronize()
n = 300
t = time.perf_counter()
for _ in range(n):
    ops.mlpg_generation(feat, var, 62, off)
print("host side: %.1f us a call (queued, no profiler)" % ((time.perf_counter() - t) / n * 1e6))
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    ops.mlpg_generation(feat, var, 62, off)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
