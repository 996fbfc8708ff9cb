"""batch_pad_gather / batch_pack_rows: dword lanes (row pitch not a multiple of 16 bytes: 425 / 187 floats) against float4
lanes (pitch 428 / 188) on a mini-batch of the bench's shape (32 utterances of 2-10 s).  HIP events over 50 launches."""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from idiaptts_amd import ops

dev = torch.device("cuda", 0)
rng = np.random.default_rng(3)
lens = rng.integers(400, 2000, size=32)
starts = np.concatenate([[0], np.cumsum(lens)[:-1]])
n, t_max = int(lens.sum()), int(lens.max())
d_starts = torch.from_numpy(starts).to(dev)
d_lens = torch.from_numpy(lens).to(dev)


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for width in (425, 187):
    pitch = (width + 3) // 4 * 4
    arena = torch.zeros((n + 8, pitch), device=dev)
    arena[:, :width] = torch.randn((n + 8, width), device=dev)
    narrow = arena[:, :width].contiguous()
    t_plain = timed(lambda: ops.batch_pad_gather(narrow, d_starts, d_lens, 32, t_max, False))
    t_vec = timed(lambda: ops.batch_pad_gather(arena, d_starts, d_lens, 32, t_max, False))
    p_plain, _ = ops.batch_pad_gather(narrow, d_starts, d_lens, 32, t_max, False)
    p_vec, _ = ops.batch_pad_gather(arena, d_starts, d_lens, 32, t_max, False)
    assert torch.equal(p_plain, p_vec[:, :, :width])
    k_plain = timed(lambda: ops.batch_pack_rows(p_plain, d_starts, d_lens, False, n + 1, out_width=pitch))
    k_vec = timed(lambda: ops.batch_pack_rows(p_vec, d_starts, d_lens, False, n + 1, out_width=pitch))
    mb = (n * width + 32 * t_max * width) * 4 / 1e6
    print("width %d (%d rows -> %d positions, %.0f MB moved): gather %.1f us dword lanes, %.1f us float4 (pitch %d); "
          "pack %.1f / %.1f us" % (width, n, 32 * t_max, mb, t_plain, t_vec, pitch, k_plain, k_vec))
