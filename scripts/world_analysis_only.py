"""The bench's WORLD analysis section alone (256 utterances, 16 kHz by default), N passes: for
rocprofv3 --kernel-trace --stats (per-kernel totals / passes = time per analysis).
usage (GPU box): python3 scripts/world_analysis_only.py [utts] [fs] [passes]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from idiaptts_amd import lib, ops, world                      # noqa: E402
from idiaptts_amd.bench_support import make_audio_batch        # noqa: E402

n_utts = int(sys.argv[1]) if len(sys.argv) > 1 else 256
fs = int(sys.argv[2]) if len(sys.argv) > 2 else 16000
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = torch.device("cuda", 0)
L = lib.load()
raws = make_audio_batch(n_utts, fs, seed=0)
hop = 5.0
order, alpha = 59, L.itts_mcep_alpha(fs)
n_fft = L.itts_cheaptrick_fft_size(fs, 71.0)
x_off = world.offsets([len(r) for r in raws])
f_off = world.offsets([world.num_frames(len(r), fs, hop) for r in raws])
x = torch.from_numpy(np.concatenate(raws)).to(dev)
stream = torch.cuda.current_stream()
side = world._side_stream(dev)


def analysis():
    f0 = ops.stonemask(x, x_off, ops.dio(x, x_off, f_off, fs, hop), f_off, fs, hop)
    if os.environ.get("SERIAL"):       # one stream: per-kernel durations mean something
        _, bap = ops.d4c(x, x_off, f0, f_off, fs, hop, n_fft, want_ap=False, want_bap=torch.float32)
    else:
        side.wait_stream(stream)
        with torch.cuda.stream(side):
            _, bap = ops.d4c(x, x_off, f0, f_off, fs, hop, n_fft, want_ap=False, want_bap=torch.float32)
    _, mc, it = ops.cheaptrick_mcep(x, x_off, f0, f_off, fs, hop, n_fft, want_sp=False, order=order,
                                    alpha=alpha, want_iters=True)
    stream.wait_stream(side)
    return f0, mc, bap, it


analysis()
torch.cuda.synchronize()
ts = []
for _ in range(passes):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    analysis()
    b.record()
    torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
print("analysis of %d utterances at %d Hz: %s ms" % (n_utts, fs, ", ".join("%.2f" % t for t in ts)))
