"""Would the backward of the recurrent stack gain from running the weight-gradient products of all
layers at the end (fewer GEMM milliseconds inside the slow stretch behind a recurrence kernel, DESIGN
12g)?  Stand-in sequence with the real kernels: P = a persistent BiLSTM layer (64 utterances, 16 input
columns: its projection is negligible), dW / dWhh / dX = the layer's products at 73 138 rows.
  order A (now):      3 x [P, dW, dWhh, dWhh, dX]
  order B (deferred): 3 x [P, dX], then 3 x [dW, dWhh, dWhh]
usage (GPU box): python3 scripts/defer_dw_probe.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from idiaptts_amd import ops                                  # noqa: E402
from idiaptts_amd.bench_support import utterance_lengths        # noqa: E402
from idiaptts_amd.nn import LSTM                               # noqa: E402

dev = torch.device("cuda:0")
M = 73138
x = torch.randn(M, 1024, device=dev)
dz = torch.randn(M, 4096, device=dev)
w = torch.randn(4096, 1024, device=dev) * 0.05
dw = torch.empty(4096, 1024, device=dev)
dx = torch.empty(M, 1024, device=dev)
h = torch.randn(M, 512, device=dev)
dzh = torch.randn(M, 2048, device=dev)
dwh = torch.empty(2048, 512, device=dev)
lengths = utterance_lengths(64, seed=3)
lstm = LSTM(16, 512, 1, bidirectional=True).to(dev)
inp = torch.randn(int(max(lengths)), 64, 16, device=dev)
lens = torch.tensor(lengths)


def P():
    with torch.no_grad():
        lstm(inp, None, lens)


def dW():
    ops.linear_bwd_weight(dz, x, dw=dw, want_bias=False)


def dWhh():
    ops.linear_bwd_weight(dzh, h, dw=dwh, want_bias=False)


def dX():
    ops.linear_bwd_input(dz, w, out=dx)


def order_a():
    for _ in range(3):
        P(); dW(); dWhh(); dWhh(); dX()


def order_b():
    for _ in range(3):
        P(); dX()
    for _ in range(3):
        dW(); dWhh(); dWhh()


def timed(fn, n=8):
    out = []
    for _ in range(n):
        s = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        e.synchronize()
        out.append(s.elapsed_time(e))
    return np.array(out[2:])


for name, fn in (("A (now)", order_a), ("B (deferred)", order_b), ("A (now)", order_a), ("B (deferred)", order_b)):
    t = timed(fn)
    print("order %-13s median %.2f ms  min %.2f  max %.2f" % (name, np.median(t), t.min(), t.max()))
