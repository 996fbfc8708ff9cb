"""The LDS-DMA GEMM on the recurrent layers' shapes (M = 73 138 frames), each product in a loop of its
own: TF/s per product, the forward and input-gradient products checked against torch.  (The grouped tile
order of the forward product -- nn.hip ring_group -- was chosen with this script and two lab switches,
group width and tile shape, that are not in the library any more; profiles/r4zz_ring_order.txt.)
usage (GPU box): python3 scripts/ring_order_probe.py [M]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from idiaptts_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def t(fn, n=20, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True)
    e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / n


M = int(sys.argv[1]) if len(sys.argv) > 1 else 73138
tag = "ring"
tot = 0.0
for (N, K) in ((4096, 1024), (4096, 428), (2048, 512), (512, 512)):
    x = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) * 0.05
    b = torch.randn(N, device=dev)
    o = torch.empty(M, N, device=dev)
    ms = t(lambda: ops.linear_fwd(x, w, b, 0, out=o))
    ref = x[:4096].double() @ w.double().t() + b.double()
    err = float((o[:4096].double() - ref).abs().max() / ref.abs().max())
    ref2 = x[-300:].double() @ w.double().t() + b.double()
    err = max(err, float((o[-300:].double() - ref2).abs().max() / ref2.abs().max()))
    print("%s fwd  N=%4d K=%4d %8.1f us %6.1f TF/s  err %.1e" % (tag, N, K, ms * 1e3, 2 * M * N * K / ms / 1e9, err))
    tot += ms
    dz = torch.randn(M, N, device=dev)
    dx = torch.empty(M, K, device=dev)
    ms = t(lambda: ops.linear_bwd_input(dz, w, out=dx))
    ref = dz[:2048].double() @ w.double()
    err = float((dx[:2048].double() - ref).abs().max() / ref.abs().max())
    print("%s dX   N=%4d K=%4d %8.1f us %6.1f TF/s  err %.1e" % (tag, N, K, ms * 1e3, 2 * M * N * K / ms / 1e9, err))
    tot += ms
    dw = torch.empty(N, K, device=dev)
    ms = t(lambda: ops.linear_bwd_weight(dz, x, dw=dw, want_bias=False))
    print("%s dW   N=%4d K=%4d %8.1f us %6.1f TF/s" % (tag, N, K, ms * 1e3, 2 * M * N * K / ms / 1e9))
    tot += ms
print("%s total %.2f ms" % (tag, tot))
