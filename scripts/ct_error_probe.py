"""Where the CheapTrick envelope of the GPU differs from the oracle's, and by how much: the wave kernel
(fft 1024) and the workgroup kernel (fft 2048 forced at the same sampling rate) on the fixture audio.
usage (GPU box): python scripts/ct_error_probe.py"""
import os
import sys

import numpy as np
import torch
from scipy.io import wavfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from idiaptts_amd import ops      # noqa: E402
from oracle import capi           # noqa: E402

gold = os.path.join(ROOT, "tests", "golden")
for name in ["LJ001-0008", "LJ001-0002"]:
    fs, w = wavfile.read(os.path.join(gold, name + ".wav"))
    raw = w.astype(np.float64) / 32768.0
    x = np.append(raw[0], raw[1:] - 0.97 * raw[:-1])
    f0, tp = capi.dio(x, fs)
    f0 = capi.stonemask(x, fs, tp, f0)
    for fft in (1024, 2048):
        sp, _, _ = ops.cheaptrick_mcep(torch.from_numpy(x).cuda(), [0, len(x)], torch.from_numpy(f0).cuda(),
                                       [0, len(f0)], fs, fft_size=fft)
        ref = capi.cheaptrick(x, fs, tp, f0, fft_size=fft)
        err = np.abs(sp.cpu().numpy() / ref - 1)
        t, k = np.unravel_index(np.argmax(err), err.shape)
        q = np.quantile(err, [0.5, 0.99, 0.9999])
        print("%s fft %d: max %.2e at frame %d bin %d (f0 %.1f, sp %.2e, frame max %.2e); median %.1e p99 %.1e p99.99 %.1e"
              % (name, fft, err.max(), t, k, f0[t], ref[t, k], ref[t].max(), q[0], q[1], q[2]))
