"""Persistent backward recurrence against the per-step kernels: time of forward + backward of one
bidirectional 512-unit layer at the bench's batch (a build with -DPERSIST_BWD_TRACE=1 prints where a
step's time goes).  usage (GPU box): python scripts/exp_rnn_persist_bwd.py [LSTM|GRU]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from idiaptts_amd import nn as inn
from idiaptts_amd.bench_support import utterance_lengths

dev = torch.device("cuda", 0)
cell = sys.argv[1] if len(sys.argv) > 1 else "LSTM"
for B, seed in ((64, 5),):
    lens = torch.from_numpy(utterance_lengths(B, seed=seed).astype(np.int64))
    T = int(lens.max())
    torch.manual_seed(B)
    layer = getattr(inn, cell)(1024, 512, 1, bidirectional=True).to(dev)
    x = torch.randn(T, B, 1024, device=dev, requires_grad=True)
    w = torch.randn(T, B, 1024, device=dev) / 8

    def run(mode, n=3):
        os.environ["ITTS_RNN_PERSISTENT_BWD"] = mode
        ts = []
        for _ in range(n + 1):
            out, _ = layer(x, None, lens)
            loss = (out * w).sum()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            g = torch.autograd.grad(loss, [x] + list(layer.parameters()))
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        return g, sorted(ts[1:])[n // 2]

    g0, t0 = run("0")
    g1, t1 = run("1", n=1 if os.environ.get("ONCE") else 3)
    d = max(float((p - q).abs().max() / max(1.0, float(q.abs().max()))) for p, q in zip(g1, g0))
    print("%s B %d T %d: backward of the layer, per-step %.2f ms, persistent %.2f ms; worst relative "
          "gradient difference %.2e" % (cell, B, T, t0, t1, d))
