"""Per-phase timeline of the MLPG scan kernel (ITTS_MLPG_SCAN_TRACE stamps, 100 MHz wall clock)."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from idiaptts_amd import ops, world
from idiaptts_amd.bench_support import utterance_lengths

dev = torch.device("cuda:0")
os.environ["ITTS_MLPG_PATH"] = sys.argv[1] if len(sys.argv) > 1 else "stream"
off = world.offsets(utterance_lengths(int(os.environ.get("MLPG_UTTS", "256")), seed=5).tolist())
feat = torch.randn(off[-1], 186, dtype=torch.float64, device=dev)
var = torch.rand(186, dtype=torch.float64, device=dev) * 0.99 + 0.01
for _ in range(3):
    ops.mlpg_generation(feat, var, 62, off)
torch.cuda.synchronize()
os.environ["ITTS_MLPG_SCAN_TRACE"] = "/tmp/mlpg_scan_trace.txt"
ops.mlpg_generation(feat, var, 62, off)
torch.cuda.synchronize()
t = np.loadtxt("/tmp/mlpg_scan_trace.txt", dtype=np.int64)
st = t[:, 2:].astype(np.float64)
st = st[st[:, 0] > 0]
us = (st - st[:, 0].min()) / 100.0
names = ["setup+mats", "fwd fold", "barrier", "fwd walk", "bwd fold", "barriers", "bwd walk"]
print("waves", len(us), "span %.1f us; start p0/p50/p100 %s" % (us[:, 7].max(), np.percentile(us[:, 0], [0, 50, 100]).round(1)))
d = np.diff(us, axis=1)
for i, n in enumerate(names):
    print("%-12s mean %6.2f p50 %6.2f p90 %6.2f max %6.2f us" % (n, d[:, i].mean(), np.median(d[:, i]), np.percentile(d[:, i], 90), d[:, i].max()))
print("wave lifetime mean %.1f max %.1f" % ((us[:, 7] - us[:, 0]).mean(), (us[:, 7] - us[:, 0]).max()))
w = t[:, 1][t[:, 2] > 0]
for ww in (0, 1, 2, 3, 8, 14, 15):
    m = w == ww
    print("wave %2d: " % ww + " ".join("%6.2f" % x for x in d[m].mean(axis=0)))
