"""Does MLPG gain from solving the batch as k independent parts on k streams (the latency-bound scan
and prep kernels of one part beside the bandwidth-bound reduce / solve kernels of another)?
usage (GPU box): python scripts/exp_mlpg_split.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from idiaptts_amd import ops, world                     # noqa: E402
from idiaptts_amd.bench_support import utterance_lengths  # noqa: E402

dev = torch.device("cuda", 0)
var = torch.rand(186, dtype=torch.float64, device=dev) * 0.99 + 0.01
streams = [torch.cuda.Stream(device=dev) for _ in range(3)]


def timed(fn, n=9):
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


for n_u in (256, 1024):
    off = world.offsets(utterance_lengths(n_u, seed=5).tolist())
    fr = off[-1]
    feat = torch.randn(fr, 186, dtype=torch.float64, device=dev)
    out = torch.empty(fr, 62, dtype=torch.float64, device=dev)
    ref = ops.mlpg_generation(feat, var, 62, off)
    for parts in (1, 2, 3, 4):
        cuts = [round(i * n_u / parts) for i in range(parts + 1)]

        def run():
            main = torch.cuda.current_stream()
            for p in range(parts):
                u0, u1 = cuts[p], cuts[p + 1]
                sub = [o - off[u0] for o in off[u0:u1 + 1]]
                st = main if p == 0 else streams[p - 1]
                if p:
                    st.wait_stream(main)
                with torch.cuda.stream(st):
                    ops.mlpg_generation(feat[off[u0]:off[u1]], var, 62, sub, out=out[off[u0]:off[u1]])
            for p in range(1, parts):
                main.wait_stream(streams[p - 1])

        ms = timed(run)
        print("%5d utts, %d part(s): %.3f ms (%.1f %% of 8 TB/s)  max |diff| %.1e" % (
            n_u, parts, ms, fr * 2000 / ms / 1e6 / 80.0, float((out - ref).abs().max())), flush=True)
