"""How long does one weight-gradient GEMM ([4096 x 1024], K = 73 138 rows) take right behind (a) the
same GEMM, (b) 8 ms of an almost idle chip (one spinning workgroup), (c) a persistent bidirectional
LSTM layer (64 utterances, the bench's lengths)?  Each case 12 times, HIP events around the GEMM only.
usage (GPU box): python3 scripts/after_what_probe.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from idiaptts_amd import ops                                  # noqa: E402
from idiaptts_amd.bench_support import utterance_lengths        # noqa: E402
from idiaptts_amd.nn import LSTM                               # noqa: E402

dev = torch.device("cuda:0")
M, N, K = 73138, 4096, 1024
x = torch.randn(M, K, device=dev)
dz = torch.randn(M, N, device=dev)
dw = torch.empty(N, K, device=dev)
w = torch.randn(N, K, device=dev) * 0.05
o = torch.empty(M, N, device=dev)


def gemm_dw():
    ops.linear_bwd_weight(dz, x, dw=dw, want_bias=False)


def gemm_fwd():
    ops.linear_fwd(x, w, None, 0, out=o)


lengths = utterance_lengths(64, seed=3)
T = int(max(lengths))
lstm = LSTM(512, 512, 1, bidirectional=True).to(dev)
inp = torch.randn(T, 64, 512, device=dev)
lens = torch.tensor(lengths)


def persistent_layer():
    with torch.no_grad():
        lstm(inp, None, lens)


def spin():
    torch.cuda._sleep(int(8e-3 * 100e6 * 21))      # ~8 ms (the counter runs at about 2.1 GHz here)


def timed(before, gemm, n=12):
    out = []
    for _ in range(n):
        before()
        s = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        s.record()
        gemm()
        e.record()
        e.synchronize()
        out.append(s.elapsed_time(e))
    return np.array(out[2:])


for _ in range(30):
    gemm_dw()
torch.cuda.synchronize()
for name, gemm in (("dW", gemm_dw), ("fwd", gemm_fwd)):
    for label, before in (("the same GEMM", gemm), ("8 ms spin of one workgroup", spin),
                          ("a persistent BiLSTM layer", persistent_layer),
                          ("host idle 20 ms", lambda: (torch.cuda.synchronize(), __import__("time").sleep(0.02)))):
        t = timed(before, gemm)
        print("%-4s behind %-28s median %.3f ms  min %.3f  max %.3f" % (name, label, np.median(t), t.min(), t.max()))
