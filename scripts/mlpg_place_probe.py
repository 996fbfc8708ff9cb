"""Does the time of the one-pass MLPG solve depend on WHERE its buffers lie?  Same process, same data, the buffers
re-allocated behind pads of different sizes; prints addresses and medians.

usage: python3 scripts/mlpg_place_probe.py [utterances] [f32|f64] [trials]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from idiaptts_amd import ops, world                             # noqa: E402
from idiaptts_amd.bench_support import utterance_lengths       # noqa: E402

n_utts = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
f32 = len(sys.argv) > 2 and sys.argv[2] == "f32"
trials = int(sys.argv[3]) if len(sys.argv) > 3 else 8
dev = torch.device("cuda", 0)
dim = 62
off = world.offsets(utterance_lengths(n_utts, seed=5).tolist())
var = torch.rand(3 * dim, dtype=torch.float64, device=dev) * 0.99 + 0.01
passes = 60 if n_utts <= 512 else 16


def measure(feat):
    t = time.time()
    while time.time() - t < 0.3:
        for _ in range(3):
            out = ops.mlpg_generation(feat, var, dim, off)
        torch.cuda.synchronize()
    ts = []
    for _ in range(passes):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = ops.mlpg_generation(feat, var, dim, off)
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0], out.data_ptr()


g = torch.Generator(device=dev)
for k in range(trials):
    torch.cuda.empty_cache()
    pad = torch.empty((k * 37 + 1) * 1234567, dtype=torch.uint8, device=dev) if k else None
    g.manual_seed(1)
    feat = torch.randn(off[-1], 3 * dim, dtype=torch.float32 if f32 else torch.float64, device=dev, generator=g)
    med, mn, optr = measure(feat)
    print("trial %d  feat %#x  out %#x  median %.1f us  min %.1f us" % (k, feat.data_ptr(), optr, med, mn), flush=True)
    del feat, pad
# the same buffers again and again: is a placement's time stable?
g.manual_seed(1)
feat = torch.randn(off[-1], 3 * dim, dtype=torch.float32 if f32 else torch.float64, device=dev, generator=g)
for k in range(3):
    med, mn, optr = measure(feat)
    print("same buffers, round %d  feat %#x  out %#x  median %.1f us  min %.1f us" % (k, feat.data_ptr(), optr, med, mn), flush=True)
