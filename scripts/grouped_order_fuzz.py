"""Random shapes for the forward product whose weights are above ~4 MB (grouped tile order, nn.hip ring_group):
every output element against torch fp64; bias + tanh / none; strided input rows.
usage (GPU box): python3 scripts/grouped_order_fuzz.py [cases]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from idiaptts_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
rng = np.random.default_rng(11)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
worst = 0.0
for case in range(n):
    M = int(rng.integers(1, 3000))
    N = int(rng.integers(1030, 5000))
    K = int(rng.integers(75, 525)) * 4
    while N * K * 4 < (5 << 20):
        N += 512
    act = int(rng.integers(0, 2))
    pad = int(rng.integers(0, 3)) * 4
    xb = torch.rand(M, K + pad, device=dev)
    x = xb[:, :K]
    w = (torch.rand(N, K, device=dev) - 0.5) * (2.0 / K ** 0.5)
    b = torch.rand(N, device=dev) - 0.5
    z = torch.nn.functional.linear(x.double(), w.double(), b.double())
    ref = torch.tanh(z) if act == 1 else z
    y = ops.linear_fwd(x, w, b, act).double()
    err = float((y - ref).abs().max())
    assert err < 2e-5, (case, M, N, K, act, err)
    worst = max(worst, err)
print("cases %d worst abs error %.2e" % (n, worst))
