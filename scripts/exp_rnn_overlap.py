"""Does a weight-gradient GEMM on a side stream run under a recurrence (step kernels, main stream)?
Measured (MI355X, round 3): no -- recurrence 22.4 ms, GEMM 5.25 ms, together 25.9 ms; with three
GEMMs 36.6 ms against 38.2 ms one after the other.  The step kernels slow down by almost the
GEMM's duration while it is resident.  usage (GPU box): python scripts/exp_rnn_overlap.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from idiaptts_amd import lib, ops
from idiaptts_amd import nn as inn
from idiaptts_amd.bench_support import utterance_lengths

dev = torch.device("cuda", 0)
L = lib.load()
lens = torch.from_numpy(np.sort(utterance_lengths(64, seed=5))[::-1].copy().astype(np.int64))
T, B, H = int(lens.max()), 64, 512
N = int(lens.sum())
layer = inn.LSTM(1024, H, 1, bidirectional=True).to(dev)
x = torch.randn(T, B, 1024, device=dev)
dg = torch.randn(N, 4096, device=dev)
xin = torch.randn(N, 1024, device=dev)
side = torch.cuda.Stream()


def recurrence():
    with torch.no_grad():
        layer(x, None, lens)


def gemm(bg):
    ops.linear_bwd_weight(dg, xin)


def timed(fn, n=3):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        fn()
        side.synchronize()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


def both(bg, n_gemm):
    def f():
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(n_gemm):
                gemm(bg)
        recurrence()
        torch.cuda.current_stream().wait_stream(side)
    return f


print("recurrence alone (gin GEMM + %d forward steps): %.2f ms" % (T, timed(recurrence)))
print("weight-gradient GEMM alone: %.2f ms" % timed(lambda: gemm(False)))
for n_gemm in (1, 2, 3):
    print("%d GEMM(s) on the side stream + recurrence: %.2f ms" % (n_gemm, timed(both(False, n_gemm))))
