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
os.environ["ITTS_MLPG_PATH"] = "fused"
os.environ["ITTS_MLPG_GEOM"] = sys.argv[1] if len(sys.argv) > 1 else "8x16x4"
N_UTT = int(os.environ.get("MLPG_UTTS", "256"))
ml_off = world.offsets(utterance_lengths(N_UTT, seed=5).tolist())
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
# steady state: waves that start in the middle half of the kernel
mid = (us[:, 0] > 0.25 * us[:, 7].max()) & (us[:, 0] < 0.75 * us[:, 7].max())
if mid.sum() > 100:
    dm = d[mid]
    print("-- middle half of the kernel (%d waves)" % mid.sum())
    for i, nme in enumerate(names[1:]):
        print("%-14s mean %7.2f  p50 %7.2f  p90 %7.2f us" % (
            nme, dm[:, i].mean(), np.median(dm[:, i]), np.percentile(dm[:, i], 90)))
    life = (us[:, 7] - us[:, 0])[mid]
    print("lifetime mean %.1f p50 %.1f p90 %.1f us" % (life.mean(), np.median(life), np.percentile(life, 90)))
    # concurrently live waves at the kernel's midpoint
    tm = 0.5 * us[:, 7].max()
    print("waves alive at the midpoint:", int(((us[:, 0] <= tm) & (us[:, 7] > tm)).sum()))
# how long after the LAST predecessor published (its stamp 3, wave 0) does a super-chunk get past
# its forward wait (stamp 4)?  Separates "the predecessor is late" from "its stores become
# visible late".
geom = os.environ["ITTS_MLPG_GEOM"].split("x")
FL, FW = int(geom[0]), int(geom[1])
full = (t[:, 2:].astype(np.float64) - t0) / 100.0
full = full.reshape(-1, FW, 8)
lens = np.diff(np.asarray(ml_off))
first_sc = []
n = 0
for T in lens:
    K = (int(T) + FL - 1) // FL
    m = (K + FW - 1) // FW
    first_sc.append((n, m))
    n += m
lat, late, startgap = [], [], []
for (f0, m) in first_sc:
    for s_ in range(f0 + 1, f0 + m):
        pub = full[f0:s_, 0, 3].max()
        got = full[s_, 0, 4]
        began = full[s_, 0, 3]
        lat.append(got - pub)
        late.append(pub - began)
        startgap.append(full[s_, 0, 0] - full[s_ - 1, 0, 0])
lat, late, startgap = np.asarray(lat), np.asarray(late), np.asarray(startgap)
print("super-chunks with predecessors:", len(lat))
print("last predecessor's publish -> past the wait: p10 %.2f p50 %.2f p90 %.2f us" % tuple(np.percentile(lat, [10, 50, 90])))
print("last predecessor's publish minus start of waiting: p10 %.2f p50 %.2f p90 %.2f us" % tuple(np.percentile(late, [10, 50, 90])))
print("start of super-chunk minus start of the one before: p10 %.2f p50 %.2f p90 %.2f us" % tuple(np.percentile(startgap, [10, 50, 90])))
