"""Harvest timing on bench.py's analysis workload (utterances of synthetic speech at 16 kHz).
usage: python3 scripts/bench_harvest.py [n_utts] [seconds]"""
import os
import sys
import time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from idiaptts_amd import ops
from idiaptts_amd.synthetic_audio import make_audio

n_utts = int(sys.argv[1]) if len(sys.argv) > 1 else 64
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 6.6
fs = 16000
dev = torch.device("cuda:0")
xs = [make_audio(fs, seconds * (0.6 + 0.8 * ((7 * k) % 10) / 10.0), k) for k in range(n_utts)]
x_off = np.concatenate([[0], np.cumsum([len(x) for x in xs])]).tolist()
T = [ops.harvest_num_frames(len(x), fs, 5.0) for x in xs]
f_off = np.concatenate([[0], np.cumsum(T)]).tolist()
x = torch.from_numpy(np.concatenate(xs)).to(dev)
audio_s = x_off[-1] / fs
f0 = ops.harvest(x, x_off, f_off, fs)
torch.cuda.synchronize()
ts = []
for _ in range(3):
    t0 = time.perf_counter()
    f0 = ops.harvest(x, x_off, f_off, fs)
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
ms = 1e3 * float(np.median(ts))
print("harvest: %d utterances, %.1f s of audio: %.1f ms  (RTF %.2e, %.0f x real time), voiced %.2f" % (
    n_utts, audio_s, ms, ms / 1e3 / audio_s, audio_s / (ms / 1e3), float((f0 > 0).double().mean())))
t0 = time.perf_counter()
d = ops.dio(x, x_off, f_off, fs)
d = ops.stonemask(x, x_off, d, f_off, fs)
torch.cuda.synchronize()
t0 = time.perf_counter()
d = ops.dio(x, x_off, f_off, fs)
d = ops.stonemask(x, x_off, d, f_off, fs)
torch.cuda.synchronize()
print("dio + stonemask on the same audio: %.1f ms, voiced %.2f" % (1e3 * (time.perf_counter() - t0),
                                                                float((d > 0).double().mean())))
if os.environ.get("HARVEST_CPU"):
    from oracle import capi
    t0 = time.perf_counter()
    capi.harvest(xs[0], fs)
    dt = time.perf_counter() - t0
    print("oracle (1 core) on utterance 0 (%.1f s): %.2f s -> RTF %.3f" % (len(xs[0]) / fs, dt, dt * fs / len(xs[0])))
