"""MLPG solve time against batch size.
Usage (GPU box): python scripts/mlpg_curve.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from idiaptts_amd import ops, world                     # noqa: E402
from idiaptts_amd.bench_support import utterance_lengths  # noqa: E402


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


dev = torch.device("cuda", 0)
var = torch.rand(186, dtype=torch.float64, device=dev) * 0.99 + 0.01
for n_u in [int(v) for v in os.environ.get("MLPG_SIZES", "256,1024,4096").split(",")]:
    off = world.offsets(utterance_lengths(n_u, seed=5).tolist())
    fr = off[-1]
    feat = torch.randn(fr, 186, dtype=torch.float64, device=dev)
    ms = timed(lambda: ops.mlpg_generation(feat, var, 62, off))
    print("%5d utts %8d frames %8.3f ms  %6.1f GB/s algorithmic (%.1f %% of 8 TB/s)" % (
        n_u, fr, ms, fr * 2000 / ms / 1e6, fr * 2000 / ms / 1e6 / 80.0), flush=True)
    del feat
