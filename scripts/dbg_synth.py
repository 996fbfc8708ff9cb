import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scipy.io import wavfile
from idiaptts_amd import ops
from oracle import capi
fs, w = wavfile.read("tests/golden/LJ001-0008.wav"); raw = w.astype(np.float64)/32768.0
f0, sp, ap = capi.wav2world(raw, fs)
ref = capi.synthesize(f0, sp, ap, fs)
dev = torch.device("cuda:0")
y,_ = ops.world_synthesize(torch.from_numpy(f0).to(dev), torch.from_numpy(sp).to(dev), torch.from_numpy(ap).to(dev), [0,len(f0)], fs, dtype=torch.float64)
y = y.cpu().numpy(); r32 = ref.astype(np.float32).astype(np.float64)
d = y - r32
print("rms ref", np.sqrt(np.mean(ref**2)), "rmse", np.sqrt(np.mean(d**2)), "max", np.abs(d).max(), "argmax", np.abs(d).argmax(), len(y))
blk = 800
e = np.array([np.sqrt(np.mean(d[i:i+blk]**2)) for i in range(0, len(d), blk)])
print(np.array2string(e, precision=1, max_line_width=200))
# unvoiced-only and voiced-only variants
f0z = np.zeros_like(f0)
refz = capi.synthesize(f0z, sp, ap, fs)
yz,_ = ops.world_synthesize(torch.from_numpy(f0z).to(dev), torch.from_numpy(sp).to(dev), torch.from_numpy(ap).to(dev), [0,len(f0)], fs, dtype=torch.float64)
print("all-unvoiced rmse", np.sqrt(np.mean((yz.cpu().numpy()-refz.astype(np.float32))**2)))
