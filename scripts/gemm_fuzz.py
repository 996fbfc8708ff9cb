"""Random shapes through the fp32 GEMM entry points (forward with bias + activation, fused backward,
loss epilogue) against torch in float64.  usage (GPU box): python scripts/gemm_fuzz.py [cases] [seed]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from idiaptts_amd import ops  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
dev = torch.device("cuda", 0)
worst = {"fwd": 0.0, "dw": 0.0, "db": 0.0, "dx": 0.0, "loss": 0.0, "dz": 0.0}


def padded(rows, cols, pad):
    pitch = (cols + 3) // 4 * 4 if pad else cols
    return torch.zeros(rows, pitch, device=dev)[:, :cols]


for case in range(n_cases):
    big = rng.random() < 0.2
    M = int(rng.integers(1, 40000 if big else 3000))
    N = int(rng.choice([1, 3, 16, 64, 65, 127, 128, 187, 256, 512, int(rng.integers(1, 700))]))
    K = int(rng.choice([1, 4, 31, 32, 33, 64, 96, 425, 512, int(rng.integers(1, 700))]))
    pad = rng.random() < 0.8
    act = int(rng.integers(0, 3))
    g = torch.Generator().manual_seed(seed * 100003 + case)
    x = padded(M, K, pad); x.copy_(torch.tanh(torch.randn(M, K, generator=g)))
    w = padded(N, K, pad); w.copy_(torch.randn(N, K, generator=g) * 0.1)
    w = w if w.is_contiguous() else None
    if w is None:      # the weight must be contiguous: keep the padded pitch inside the matrix
        Kp = (K + 3) // 4 * 4
        w = torch.zeros(N, Kp, device=dev)
        w[:, :K] = torch.randn(N, K, generator=g).to(dev) * 0.1
        xx = torch.zeros(M, Kp, device=dev); xx[:, :K] = x; x = xx
        K = Kp
    b = torch.randn(N, generator=g).to(dev)
    xd, wd, bd = x.double().cpu(), w.double().cpu(), b.double().cpu()
    z = xd @ wd.t() + bd
    y_ref = z if act == 0 else (torch.tanh(z) if act == 1 else torch.relu(z))
    y = ops.linear_fwd(x, w, b, act, out=padded(M, N, pad))
    tol = 3e-5 * max(1.0, float(z.abs().max()))
    e = float((y.double().cpu() - y_ref).abs().max())
    assert e < tol, ("fwd", case, M, N, K, pad, act, e)
    worst["fwd"] = max(worst["fwd"], e / tol)
    # fused backward (previous layer: tanh of x)
    dz = padded(M, N, pad); dz.copy_(torch.randn(M, N, generator=g))
    dw = torch.zeros(N, K, device=dev); db = torch.zeros(N, device=dev); dx = padded(M, K, pad)
    ops.linear_bwd(dz, x, w, dw, db, dx, yprev=x, act_prev=ops.ACT_TANH)
    dzd = dz.double().cpu()
    dw_ref, db_ref = dzd.t() @ xd, dzd.sum(0)
    dx_ref = (dzd @ wd) * (1 - xd ** 2)
    for name, got, ref, rel in (("dw", dw, dw_ref, 3e-4), ("db", db, db_ref, 3e-4), ("dx", dx, dx_ref, 3e-5)):
        t = rel * max(1.0, float(ref.abs().max()))
        e = float((got.double().cpu() - ref).abs().max())
        assert e < t, (name, case, M, N, K, pad, e, t)
        worst[name] = max(worst[name], e / t)
    # output layer + masked MSE in one launch
    if M >= 2 and pad and x.stride(0) % 4 == 0 and w.shape[1] % 4 == 0:   # the fused loss needs 16-byte rows
        target = padded(M, N, pad); target.copy_(torch.randn(M, N, generator=g))
        valid = (torch.rand(M, generator=g) > 0.2).to(torch.uint8)
        valid[0] = 1
        nv = float(valid.sum())
        loss, dzo = ops.linear_fwd_mse(x, w, b, target, valid.to(dev), nv, grad=padded(M, N, pad))
        diff = (z - target.double().cpu()) * valid.double()[:, None]
        loss_ref = float((diff ** 2).sum() / (nv * N))
        dz_ref = 2 * diff / (nv * N)
        e = abs(float(loss) - loss_ref) / max(1e-12, abs(loss_ref))
        assert e < 2e-5, ("loss", case, M, N, K, pad, float(loss), loss_ref)
        worst["loss"] = max(worst["loss"], e / 2e-5)
        t = 3e-5 * max(1e-12, float(dz_ref.abs().max()))
        e = float((dzo.double().cpu() - dz_ref).abs().max())
        assert e < t, ("dz", case, M, N, K, pad, e, t)
        worst["dz"] = max(worst["dz"], e / t)
torch.cuda.synchronize()
print("cases", n_cases, "worst error as a fraction of its tolerance:", {k: round(v, 3) for k, v in worst.items()})
