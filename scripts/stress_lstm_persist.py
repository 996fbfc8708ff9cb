"""Stress of the persistent LSTM / GRU recurrences (forward: both; backward: LSTM): random ragged batches,
outputs, final states and every gradient of each launch compared with the step kernels
(ITTS_RNN_PERSISTENT=0).  usage (GPU box): python scripts/stress_lstm_persist.py [launches]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from idiaptts_amd import nn as inn

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda", 0)
rng = np.random.default_rng(3)
worst = 0.0
for it in range(n):
    bidir = bool(rng.integers(0, 2))
    B = int(rng.integers(1, 200))
    T = int(rng.choice([1, 2, 17, 100, 400, 1200]))
    lens = rng.integers(1, T + 1, size=B)
    lens[rng.integers(0, B)] = T
    lens = torch.from_numpy(lens.astype(np.int64))
    torch.manual_seed(it)
    in_dim = int(rng.choice([64, 425, 1024]))
    cell = "LSTM" if rng.random() < 0.5 else "GRU"
    layer = getattr(inn, cell)(in_dim, 512, 1, bidirectional=bidir).to(dev)
    x = torch.randn(T, B, in_dim, device=dev, requires_grad=True)
    w_out = torch.randn(T, B, 512 * (2 if bidir else 1), device=dev) / 8
    outs = []
    for mode in ("0", "1"):
        os.environ["ITTS_RNN_PERSISTENT"] = mode
        o, st = layer(x, None, lens)
        st = tuple(st) if isinstance(st, (tuple, list)) else (st,)
        loss = (o * w_out).sum()
        params = [x] + list(layer.parameters())
        grads = torch.autograd.grad(loss, params)
        outs.append((o.detach(),) + tuple(q.detach() for q in st) + tuple(grads))
    torch.cuda.synchronize()
    d = max(float((a - b).abs().max() / max(1.0, float(b.abs().max()))) for a, b in zip(outs[0], outs[1]))
    assert d < 2e-5, (it, cell, B, T, bidir, d)
    worst = max(worst, d)
print("launches", n, "worst relative difference to the step kernels %.2e" % worst)
