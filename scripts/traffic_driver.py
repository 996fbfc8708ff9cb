"""One bench section, `passes` times, and nothing else behind the set-up: the target of the PMC passes
of scripts/section_traffic.sh (HBM bytes of a pass = (counters at 3 passes - counters at 1 pass) / 2).
usage: python3 scripts/traffic_driver.py <analysis|synthesis|bilstm|bigru|mlpg> <passes> [fs] [utts]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from idiaptts_amd import lib, ops, world                      # noqa: E402
from idiaptts_amd.bench_support import make_audio_batch, utterance_lengths   # noqa: E402

section, passes = sys.argv[1], int(sys.argv[2])
fs = int(sys.argv[3]) if len(sys.argv) > 3 else 16000
n_utts = int(sys.argv[4]) if len(sys.argv) > 4 else (256 if fs <= 24000 else 64)
dev = torch.device("cuda", 0)
L = lib.load()

if section in ("analysis", "synthesis"):
    raws = make_audio_batch(n_utts, fs, seed=0)
    order, alpha = 59, L.itts_mcep_alpha(fs)
    n_fft = L.itts_cheaptrick_fft_size(fs, 71.0)
    x_off = world.offsets([len(r) for r in raws])
    f_off = world.offsets([world.num_frames(len(r), fs, 5.0) for r in raws])
    x = torch.from_numpy(np.concatenate(raws)).to(dev)

    def analysis():
        f0 = ops.stonemask(x, x_off, ops.dio(x, x_off, f_off, fs, 5.0), f_off, fs, 5.0)
        _, bap = ops.d4c(x, x_off, f0, f_off, fs, 5.0, n_fft, want_ap=False, want_bap=torch.float32)
        _, mc, _ = ops.cheaptrick_mcep(x, x_off, f0, f_off, fs, 5.0, n_fft, want_sp=False, order=order,
                                       alpha=alpha, want_iters=True)
        return f0, mc, bap

    if section == "analysis":
        for _ in range(passes):
            analysis()
    else:
        # the set-up (one analysis) is the same whatever `passes` is: it cancels in the difference
        f0, mc, bap = analysis()
        mc64, bap64 = mc.double(), bap.double()
        for _ in range(passes):          # (as the bench runs it: spectra on the side stream, voiced-only decode)
            world.synthesise_features(f0, f_off, fs, n_fft, mc=mc64, alpha=alpha, bap=bap64)
elif section in ("bilstm", "bigru"):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    bench.bilstm_section(dev, 64, max(passes - 2, 1), cell="LSTM" if section == "bilstm" else "GRU")
elif section == "mlpg":
    off = world.offsets(utterance_lengths(256, seed=5).tolist())
    feat = torch.randn(off[-1], 186, dtype=torch.float64, device=dev)
    var = torch.rand(186, dtype=torch.float64, device=dev) * 0.99 + 0.01
    for _ in range(passes):
        ops.mlpg_generation(feat, var, 62, off)
torch.cuda.synchronize()
