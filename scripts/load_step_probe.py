"""DESIGN.md section 12g said: a GEMM behind a stretch of light load (a persistent recurrence, a spinning workgroup, an
idle chip) runs 20-25 % slower for several milliseconds.  Is it the load STEP -- and would matrix work of no use during
the light stretch (in the recurrence's polling loops) prevent it?  One [4096 x 1024] weight-gradient product over 73 138
rows behind 8 ms of scripts/xcd_lab/burn.hip at several duty cycles of the matrix units (all 256 CUs, one or two
workgroups per CU), HIP events around the product only.
usage (GPU box): python3 scripts/load_step_probe.py"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from idiaptts_amd import ops                                  # noqa: E402

dev = torch.device("cuda:0")
burn = ctypes.CDLL(os.path.join(ROOT, "scripts", "xcd_lab", "libburn.so"))
burn.burn_launch.argtypes = [ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
M, N, K = 73138, 4096, 1024
x = torch.randn(M, K, device=dev)
dz = torch.randn(M, N, device=dev)
dw = torch.empty(N, K, device=dev)
sink = torch.zeros(4, device=dev)


def gemm():
    ops.linear_bwd_weight(dz, x, dw=dw, want_bias=False)


def timed(before, n=10):
    out = []
    for _ in range(n):
        before()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        gemm()
        e.record()
        e.synchronize()
        out.append(s.elapsed_time(e))
    return np.array(out[2:])


def burner(ms, on, off, wgs):
    def f():
        rc = burn.burn_launch(ms, on, off, wgs, sink.data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert rc == 0
    return f


for _ in range(30):
    gemm()
torch.cuda.synchronize()
cases = [("the same GEMM", gemm),
         ("8 ms, 1 workgroup/CU sleeping only", burner(8.0, 0, 8, 256)),
         ("8 ms, 1 wg/CU, MFMA ~1/8 duty", burner(8.0, 1, 14, 256)),
         ("8 ms, 1 wg/CU, MFMA ~1/3 duty", burner(8.0, 1, 4, 256)),
         ("8 ms, 1 wg/CU, MFMA ~1/2 duty", burner(8.0, 1, 2, 256)),
         ("8 ms, 1 wg/CU, MFMA full", burner(8.0, 4, 0, 256)),
         ("8 ms, 2 wg/CU, MFMA full", burner(8.0, 4, 0, 512)),
         ("8 ms, 1 workgroup in all", burner(8.0, 0, 8, 1))]
for label, before in cases:
    t = timed(before)
    print("dW behind %-40s median %.3f ms  min %.3f  max %.3f" % (label, np.median(t), t.min(), t.max()))
