import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from idiaptts_amd import ops
from idiaptts_amd.synthetic_audio import make_audio
from oracle import capi
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
bad = 0; tot = 0; worst = 0.0
for k in range(48):
    fs = [16000, 16000, 22050, 44100, 48000, 8000][k % 6]
    secs = float(rng.uniform(0.4, 2.5))
    x = make_audio(fs, secs, 100 + k) * float(rng.uniform(0.05, 2.0))
    if k % 5 == 0:
        x = x + 0.05 * rng.normal(size=len(x))
    fp = [5.0, 5.0, 10.0, 1.0][k % 4]
    T = ops.harvest_num_frames(len(x), fs, fp)
    f0 = ops.harvest(torch.from_numpy(x).to(dev), [0, len(x)], [0, T], fs, fp).cpu().numpy()
    ref, _ = capi.harvest(x, fs, fp)
    mism = int(((f0 > 0) != (ref > 0)).sum())
    m = (f0 > 0) & (ref > 0)
    rel = float(np.abs(f0[m] / ref[m] - 1).max()) if m.any() else 0.0
    tot += len(ref); bad += mism; worst = max(worst, rel)
    if mism or rel > 1e-7:
        print("utt", k, "fs", fs, "fp", fp, "frames", len(ref), "voicing mismatches", mism, "max rel", rel)
print("frames", tot, "voicing mismatches", bad, "worst rel", worst)
