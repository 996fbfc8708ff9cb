import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from idiaptts_amd import ops
dev = torch.device("cuda:0")
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / n
M = 73138
for (N, K) in ((4096, 1024), (4096, 428), (1024, 4096), (187, 1024)):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05; b = torch.zeros(N, device=dev)
    o = torch.empty(M, N, device=dev)
    ms = t(lambda: ops.linear_fwd(x, w, b, 0, out=o))
    print("fwd  M=%d N=%d K=%d  %8.1f us %6.1f TF/s" % (M, N, K, ms * 1e3, 2 * M * N * K / ms / 1e9))
    dz = torch.randn(M, N, device=dev); dx = torch.empty(M, K, device=dev)
    ms = t(lambda: ops.linear_bwd_input(dz, w, out=dx))
    print("dX   M=%d N=%d K=%d  %8.1f us %6.1f TF/s" % (M, N, K, ms * 1e3, 2 * M * N * K / ms / 1e9))
    dw = torch.empty(N, K, device=dev)
    ms = t(lambda: ops.linear_bwd_weight(dz, x, dw=dw, want_bias=False))
    print("dW   M=%d N=%d K=%d  %8.1f us %6.1f TF/s" % (M, N, K, ms * 1e3, 2 * M * N * K / ms / 1e9))
