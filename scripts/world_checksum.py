"""Prints SHA-1 digests of the WORLD kernels' raw fp64 outputs on fixed synthetic audio -- used to
check that a kernel restructuring (e.g. the LDS FFT's stage fusion) is bit-identical.
usage: python3 scripts/world_checksum.py"""
import hashlib
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from idiaptts_amd import ops, world
from idiaptts_amd.bench_support import make_audio_batch


def digest(t):
    return hashlib.sha1(t.detach().cpu().numpy().tobytes()).hexdigest()[:16]


for fs in (16000, 48000):
    raws = make_audio_batch(3, fs, seed=3, min_s=0.5, max_s=1.5)
    dev = torch.device("cuda")
    x_off = world.offsets([len(r) for r in raws])
    f_off = world.offsets([world.num_frames(len(r), fs, 5.0) for r in raws])
    x = torch.from_numpy(np.concatenate(raws)).to(dev)
    f0 = ops.stonemask(x, x_off, ops.dio(x, x_off, f_off, fs), f_off, fs)
    sp, mc, it = ops.cheaptrick_mcep(x, x_off, f0, f_off, fs, order=59, alpha=0.41 if fs == 16000 else 0.554,
                                     mc_dtype=torch.float64, want_iters=True)
    ap, bap = ops.d4c(x, x_off, f0, f_off, fs, want_bap=torch.float64)
    y, _ = ops.world_synthesize(f0, sp, ap, f_off, fs, dtype=torch.float64)
    pw = ops.mgc2sp(mc, 0.41 if fs == 16000 else 0.554, (sp.shape[1] - 1) * 2, want_pow=True)
    print(fs, "f0", digest(f0), "sp", digest(sp), "mc", digest(mc), "it", digest(it), "ap", digest(ap),
          "y", digest(y), "pw", digest(pw))
