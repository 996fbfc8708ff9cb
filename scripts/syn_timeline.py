"""Kernel timeline driver: synthesis as the bench runs it (world.synthesise_features), `passes` times with a
synchronise between.  usage (GPU box): rocprofv3 --kernel-trace ... -- python3 scripts/syn_timeline.py [passes]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from idiaptts_amd import lib, ops, world                      # noqa: E402
from idiaptts_amd.bench_support import make_audio_batch       # noqa: E402

passes = int(sys.argv[1]) if len(sys.argv) > 1 else 4
fs, n_utts = 16000, 256
dev = torch.device("cuda", 0)
L = lib.load()
raws = make_audio_batch(n_utts, fs, seed=0)
order, alpha = 59, L.itts_mcep_alpha(fs)
n_fft = L.itts_cheaptrick_fft_size(fs, 71.0)
x_off = world.offsets([len(r) for r in raws])
f_off = world.offsets([world.num_frames(len(r), fs, 5.0) for r in raws])
x = torch.from_numpy(np.concatenate(raws)).to(dev)
f0 = ops.stonemask(x, x_off, ops.dio(x, x_off, f_off, fs, 5.0), f_off, fs, 5.0)
_, bap = ops.d4c(x, x_off, f0, f_off, fs, 5.0, n_fft, want_ap=False, want_bap=torch.float32)
_, mc, _ = ops.cheaptrick_mcep(x, x_off, f0, f_off, fs, 5.0, n_fft, want_sp=False, order=order, alpha=alpha, want_iters=True)
mc64, bap64 = mc.double(), bap.double()
print("voiced frames: %.3f of %d" % (float((f0 > 0).double().mean()), f0.numel()))
torch.cuda.synchronize()
times = []
for _ in range(passes):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    y, y_off = world.synthesise_features(f0, f_off, fs, n_fft, mc=mc64, alpha=alpha, bap=bap64)
    e1.record()
    torch.cuda.synchronize()
    times.append(e0.elapsed_time(e1))
times.sort()
print("synthesis: median %.3f ms  min %.3f ms  (%d passes)" % (times[len(times) // 2], times[0], passes))
