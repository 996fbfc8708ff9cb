"""Synthesis only, at the bench's batch (for profiler passes): python scripts/syn_only.py [utts] [passes]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from idiaptts_amd import lib, ops, world
from idiaptts_amd.bench_support import make_audio_batch

n_utts = int(sys.argv[1]) if len(sys.argv) > 1 else 256
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 3
fs = 16000
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
_, mc, _ = ops.cheaptrick_mcep(x, x_off, f0, f_off, fs, 5.0, n_fft, want_sp=False, order=order, alpha=alpha,
                               want_iters=True)
pw = ops.mgc2sp(mc.double(), alpha, n_fft, want_pow=True)
apd = ops.decode_aperiodicity(bap.double(), fs, n_fft)
torch.cuda.synchronize()
for _ in range(passes):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    y = ops.world_synthesize(f0, pw, apd, f_off, fs, 5.0)
    b.record()
    torch.cuda.synchronize()
    print("synthesis of %d utterances: %.2f ms" % (n_utts, a.elapsed_time(b)))
y = y[0] if isinstance(y, (tuple, list)) else y
print("checksum %.10e" % float(y.double().abs().sum()))
