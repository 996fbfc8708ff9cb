"""MLPG timing: 256 utterances x 62 dims (bench.py's section), HIP events, per variant.
usage: python3 scripts/bench_mlpg.py"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from idiaptts_amd import ops, world
from idiaptts_amd.bench_support import utterance_lengths

dev = torch.device("cuda:0")
ml_off = world.offsets(utterance_lengths(256, seed=5).tolist())
n = ml_off[-1]
feat = torch.randn(n, 186, dtype=torch.float64, device=dev)
var = torch.rand(186, dtype=torch.float64, device=dev) * 0.99 + 0.01


def run(label):
    out = ops.mlpg_generation(feat, var, 62, ml_off)
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ops.mlpg_generation(feat, var, 62, ml_off)
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    ms = float(np.median(ts))
    print("%-12s %.3f ms  %.0f GB/s algorithmic (%.1f %% of 8 TB/s)" % (
        label, ms, n * 2000 / ms / 1e6, n * 2000 / ms / 1e6 / 80))
    return out


os.environ["ITTS_MLPG_MULTIPASS"] = "1"
ref = run("multipass")
os.environ["ITTS_MLPG_MULTIPASS"] = "0"
for geom in [a for a in sys.argv[1:] if "x" in a] or ["32x16", "32x8", "32x4", "32x2", "16x16", "16x8", "16x4"]:
    os.environ["ITTS_MLPG_GEOM"] = geom
    out = run("fused " + geom)
    print("   max |fused - multipass| = %.3e" % float((out - ref).abs().max()))
