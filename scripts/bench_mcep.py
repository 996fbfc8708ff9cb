import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from idiaptts_amd import ops, world, lib
from idiaptts_amd.bench_support import make_audio_batch
dev = torch.device("cuda:0")
fs = 16000
raws = make_audio_batch(16, fs, seed=0)
x_off = world.offsets([len(r) for r in raws]); f_off = world.offsets([world.num_frames(len(r), fs) for r in raws])
x = torch.from_numpy(np.concatenate(raws)).to(dev)
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / n
f0 = ops.stonemask(x, x_off, ops.dio(x, x_off, f_off, fs), f_off, fs)
T = f_off[-1]
print("frames", T)
print("dio       %.3f ms" % t(lambda: ops.dio(x, x_off, f_off, fs)))
print("stonemask %.3f ms" % t(lambda: ops.stonemask(x, x_off, f0, f_off, fs)))
print("cheaptrick sp only %.3f ms" % t(lambda: ops.cheaptrick_mcep(x, x_off, f0, f_off, fs, want_sp=True)))
sp, _, _ = ops.cheaptrick_mcep(x, x_off, f0, f_off, fs, want_sp=True)
amp = sp.sqrt()
for order in (19, 59):
    for maxiter in (1, 2, 3, 30):
        ms = t(lambda: ops.mcep(amp, order, 0.41, maxiter=maxiter, miniter=min(2, maxiter)))
        print("mcep order %d maxiter %2d: %.3f ms (%.2f us/frame-slot)" % (order, maxiter, ms, ms*1e3/T*512))
print("fused ct+mcep59 %.3f ms" % t(lambda: ops.cheaptrick_mcep(x, x_off, f0, f_off, fs, want_sp=False, order=59, alpha=0.41)))
print("d4c %.3f ms" % t(lambda: ops.d4c(x, x_off, f0, f_off, fs, want_ap=False, want_bap=torch.float32)))
