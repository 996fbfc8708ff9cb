"""Per-phase timeline of the fused MLPG kernel (ITTS_MLPG_TRACE stamps, 100 MHz wall clock).
usage: python3 scripts/mlpg_trace.py [FW]"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from idiaptts_amd import ops, world
from idiaptts_amd.bench_support import utterance_lengths

dev = torch.device("cuda:0")
os.environ["ITTS_MLPG_GEOM"] = sys.argv[1] if len(sys.argv) > 1 else "32x4"
ml_off = world.offsets(utterance_lengths(256, seed=5).tolist())
n = ml_off[-1]
feat = torch.randn(n, 186, dtype=torch.float64, device=dev)
var = torch.rand(186, dtype=torch.float64, device=dev) * 0.99 + 0.01
for _ in range(3):
    ops.mlpg_generation(feat, var, 62, ml_off)
torch.cuda.synchronize()
os.environ["ITTS_MLPG_TRACE"] = "/tmp/mlpg_trace.txt"
ops.mlpg_generation(feat, var, 62, ml_off)
torch.cuda.synchronize()
t = np.loadtxt("/tmp/mlpg_trace.txt", dtype=np.int64)
st = t[:, 2:].astype(np.float64)
ok = st[:, 0] > 0
st = st[ok]
t0 = st[:, 0].min()
us = (st - t0) / 100.0
names = ["start", "loaded", "passA_f", "folded+pub", "preds", "passB_f+A_b", "folded+pub_b", "end"]
print("waves", len(us), "kernel span %.1f us" % us[:, 7].max())
d = np.diff(us, axis=1)
for i, nme in enumerate(names[1:]):
    print("%-14s mean %7.2f  p50 %7.2f  p90 %7.2f  max %7.2f us" % (
        nme, d[:, i].mean(), np.median(d[:, i]), np.percentile(d[:, i], 90), d[:, i].max()))
print("wave lifetime mean %.1f  p90 %.1f us" % ((us[:, 7] - us[:, 0]).mean(),
                                                 np.percentile(us[:, 7] - us[:, 0], 90)))
print("start time percentiles", np.percentile(us[:, 0], [0, 25, 50, 75, 100]).round(1))
