"""Random shapes through every MLPG path against the sequential sweeps (which tests/test_gpu_mlpg.py
pins to the C oracle).  usage (GPU box): python scripts/mlpg_fuzz.py [cases] [seed]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from idiaptts_amd import ops  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device("cuda", 0)
worst = {}
for case in range(n_cases):
    n_utts = int(rng.integers(1, 24))
    kind = rng.integers(0, 4)
    hi = (40, 400, 2500, 5000)[kind]
    lengths = rng.integers(0 if kind == 0 else 1, hi, size=n_utts)
    if rng.random() < 0.3:
        lengths[rng.integers(0, n_utts)] = int(rng.choice([1, 2, 3, 15, 16, 17, 31, 32, 33, 47, 48, 49, 64, 65]))
    if lengths.sum() == 0:
        lengths[0] = 5
    dim = int(rng.choice([1, 2, 3, 7, 20, 60, 62, 64, 65, 70, 129]))
    col0, extra, ocol0, oextra = (int(rng.integers(0, 4)) for _ in range(4))
    off = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
    T = int(off[-1])
    feat = torch.from_numpy(rng.normal(size=(T, col0 + 3 * dim + extra))).to(dev)
    var = rng.uniform(0.01, 1.0, size=3 * dim)
    if rng.random() < 0.25:
        var[dim:] *= 10.0 ** rng.uniform(-6, 0)          # slowly settling factor
    var = torch.from_numpy(var).to(dev)
    outs = {}
    for path in ("seq", "stream", "direct", "stream8", "stream32", "fused", "multipass"):
        os.environ["ITTS_MLPG_PATH"] = path
        out = torch.full((T, ocol0 + dim + oextra), 3.5, dtype=torch.float64, device=dev)
        ops.mlpg_generation(feat, var, dim, off.tolist(), col0=col0, out=out, ocol0=ocol0)
        outs[path] = out
    torch.cuda.synchronize()
    ref = outs["seq"]
    scale = max(1.0, float(ref[:, ocol0:ocol0 + dim].abs().max()))
    for path, out in outs.items():
        if ocol0:
            assert bool((out[:, :ocol0] == 3.5).all()), (case, path, "left columns touched")
        if oextra:
            assert bool((out[:, ocol0 + dim:] == 3.5).all()), (case, path, "right columns touched")
        err = float((out[:, ocol0:ocol0 + dim] - ref[:, ocol0:ocol0 + dim]).abs().max()) / scale
        assert np.isfinite(err) and err < 1e-9, (case, path, err, n_utts, lengths.tolist()[:8], dim, col0, ocol0)
        worst[path] = max(worst.get(path, 0.0), err)
print("cases", n_cases, "worst relative difference to the sequential sweeps:", {k: "%.1e" % v for k, v in worst.items()})
