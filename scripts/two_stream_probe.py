"""Does the WORLD analysis gain from two half-batches in flight (two host threads, two streams: the MFMA-bound
warping GEMMs of one half beside the VALU / LDS-bound frame kernels of the other)?  Wall time of the full
batch, of the two halves one after the other, and of the two halves side by side.
usage (GPU box): python3 scripts/two_stream_probe.py [utts] [fs]"""
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from idiaptts_amd import lib, ops, world                      # noqa: E402
from idiaptts_amd.bench_support import make_audio_batch        # noqa: E402

n_utts = int(sys.argv[1]) if len(sys.argv) > 1 else 256
fs = int(sys.argv[2]) if len(sys.argv) > 2 else 16000
dev = torch.device("cuda", 0)
L = lib.load()
raws = make_audio_batch(n_utts, fs, seed=0)
hop = 5.0
order, alpha = 59, L.itts_mcep_alpha(fs)
n_fft = L.itts_cheaptrick_fft_size(fs, 71.0)


class Part(object):
    def __init__(self, rs):
        self.x_off = world.offsets([len(r) for r in rs])
        self.f_off = world.offsets([world.num_frames(len(r), fs, hop) for r in rs])
        self.x = torch.from_numpy(np.concatenate(rs)).to(dev)

    def analysis(self, serial_d4c=True):
        x, x_off, f_off = self.x, self.x_off, self.f_off
        f0 = ops.stonemask(x, x_off, ops.dio(x, x_off, f_off, fs, hop), f_off, fs, hop)
        _, bap = ops.d4c(x, x_off, f0, f_off, fs, hop, n_fft, want_ap=False, want_bap=torch.float32)
        _, mc, it = ops.cheaptrick_mcep(x, x_off, f0, f_off, fs, hop, n_fft, want_sp=False, order=order,
                                        alpha=alpha, want_iters=True)
        return mc


full = Part(raws)
halves = [Part(raws[0::2]), Part(raws[1::2])]
streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]


def wall(fn, n=4):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)[len(ts) // 2]


def sequential():
    for h in halves:
        h.analysis()


def side_by_side():
    def run(i):
        with torch.cuda.stream(streams[i]):
            halves[i].analysis()
    th = [threading.Thread(target=run, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()


print("full batch            %.2f ms" % wall(full.analysis))
print("two halves in a row   %.2f ms" % wall(sequential))
print("two halves side by side %.2f ms" % wall(side_by_side))
