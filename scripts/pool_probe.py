"""Probe of the stream-ordered scratch pool around repeated analysis passes (diagnostic)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from idiaptts_amd import lib, ops, world
from idiaptts_amd.bench_support import make_audio_batch
L = lib.load()
def stats():
    r, u, k = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)
    L.itts_scratch_pool_stats(ctypes.byref(r), ctypes.byref(u), ctypes.byref(k))
    return "reserved %.2f GB used %.2f GB keep %.0f GB" % (r.value / 2**30, u.value / 2**30, k.value / 2**30)
fs = 16000
dev = torch.device("cuda:0")
raws = make_audio_batch(int(sys.argv[1]) if len(sys.argv) > 1 else 256, fs, seed=0)
x_off = world.offsets([len(r) for r in raws]); f_off = world.offsets([world.num_frames(len(r), fs, 5.0) for r in raws])
x = torch.from_numpy(np.concatenate(raws)).to(dev)
print("start:", stats(), "free/total GB", [v / 2**30 for v in torch.cuda.mem_get_info()])
for i in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    f0 = ops.stonemask(x, x_off, ops.dio(x, x_off, f_off, fs), f_off, fs)
    _, bap = ops.d4c(x, x_off, f0, f_off, fs, 5.0, 1024, want_ap=False, want_bap=torch.float32)
    _, mc, it = ops.cheaptrick_mcep(x, x_off, f0, f_off, fs, 5.0, 1024, want_sp=False, order=59, alpha=0.41, want_iters=True)
    before = stats()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("pass %d: %.1f ms | before sync: %s | after sync: %s | free %.1f GB" % (i, dt * 1e3, before, stats(), torch.cuda.mem_get_info()[0] / 2**30))

if len(sys.argv) > 2:
    # mixed sequence: analysis / harvest on a subset / analysis ... (does a large foreign request
    # make the pool drop its cached blocks?)
    nh = 64
    xh, xoh, foh = x[:x_off[nh]], x_off[:nh + 1], [ops.harvest_num_frames(b - a, fs, 5.0) for a, b in zip(x_off[:nh], x_off[1:nh + 1])]
    foh = world.offsets(foh)
    def analysis():
        f0 = ops.stonemask(x, x_off, ops.dio(x, x_off, f_off, fs), f_off, fs)
        ops.d4c(x, x_off, f0, f_off, fs, 5.0, 1024, want_ap=False, want_bap=torch.float32)
        ops.cheaptrick_mcep(x, x_off, f0, f_off, fs, 5.0, 1024, want_sp=False, order=59, alpha=0.41)
    for i in range(12):
        which = "harvest" if i % 3 == 2 else "analysis"
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if which == "harvest":
            ops.harvest(xh, xoh, foh, fs, 5.0)
        else:
            analysis()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("%-8s %.1f ms | %s" % (which, dt * 1e3, stats()))
