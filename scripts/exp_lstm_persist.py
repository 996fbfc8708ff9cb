"""Persistent per-XCD LSTM forward (ITTS_RNN_PERSISTENT=1) against the per-step kernels: results and
time of one bidirectional 512-unit layer at the bench's batch.  usage (GPU box): python scripts/exp_lstm_persist.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from idiaptts_amd import nn as inn
from idiaptts_amd.bench_support import utterance_lengths

dev = torch.device("cuda", 0)

for B, seed in ((64, 5), (17, 3), (48, 9)):
    lens = torch.from_numpy(utterance_lengths(B, seed=seed).astype(np.int64))
    T = int(lens.max())
    torch.manual_seed(B)
    layer = inn.LSTM(1024, 512, 1, bidirectional=True).to(dev)
    x = torch.randn(T, B, 1024, device=dev)

    def run(mode, n=3):
        os.environ["ITTS_RNN_PERSISTENT"] = mode
        with torch.no_grad():
            out, (hn, cn) = layer(x, None, lens)
            torch.cuda.synchronize()
            ts = []
            for _ in range(n):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                layer(x, None, lens)
                b.record()
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(b))
        return out, hn, cn, sorted(ts)[len(ts) // 2]

    o0, h0, c0, t0 = run("0")
    o1, h1, c1, t1 = run("1")
    print("B %d T %d: per-step %.2f ms, persistent %.2f ms; max |dy| %.2e |dhn| %.2e |dcn| %.2e (|y| max %.2f)" % (
        B, T, t0, t1, float((o0 - o1).abs().max()), float((h0 - h1).abs().max()), float((c0 - c1).abs().max()),
        float(o0.abs().max())), flush=True)
