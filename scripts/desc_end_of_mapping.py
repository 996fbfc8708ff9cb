"""Weight-gradient GEMMs on column slices of a tensor that ends exactly where its mapping ends (run with
PYTORCH_NO_CUDA_MEMORY_CACHING=1: every tensor is its own hipMalloc; [4096 x 128] f32 = 2 MiB).  With
operand descriptors that ended at the last row's pitch this died with a memory access fault."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from idiaptts_amd import ops
dev = torch.device("cuda", 0)
torch.manual_seed(0)
M = 4096
for trial in range(8):
    # dz [M x 128] f32 = exactly 2 MiB, its own allocation (no caching): the tensor ends where hipMalloc's mapping ends
    dz = torch.randn(M, 128, device=dev)
    x = torch.randn(M, 16 + 4 * trial, device=dev)
    for c0 in (0, 64):
        dw, db = ops.linear_bwd_weight(dz[:, c0:c0 + 64], x)
        ref = dz[:, c0:c0 + 64].double().t() @ x.double()
        err = float((dw.double() - ref).abs().max() / ref.abs().max())
        assert err < 1e-5, err
    del dz, x
torch.cuda.synchronize()
print("ok")
