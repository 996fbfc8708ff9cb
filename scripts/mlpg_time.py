"""Time itts_mlpg_generation on the traffic_driver's MLPG batch (256 utterances, 186 -> 62 columns).

usage: python3 scripts/mlpg_time.py [passes] [utterances] [f32|f64] [dim]      ITTS_MLPG_STREAM=1 selects the three-launch form"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from idiaptts_amd import ops, world                             # noqa: E402
from idiaptts_amd.bench_support import utterance_lengths       # noqa: E402

passes = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = torch.device("cuda", 0)
torch.manual_seed(int(os.environ.get("MLPG_TIME_SEED", "0")))      # (the variances set the length of the factor's head, and that the time)
n_utts = int(sys.argv[2]) if len(sys.argv) > 2 else 256
off = world.offsets(utterance_lengths(n_utts, seed=5).tolist())
f32 = len(sys.argv) > 3 and sys.argv[3] == "f32"
dim = int(sys.argv[4]) if len(sys.argv) > 4 else 62
feat = torch.randn(off[-1], 3 * dim, dtype=torch.float32 if f32 else torch.float64, device=dev)
var = torch.rand(3 * dim, dtype=torch.float64, device=dev) * 0.99 + 0.01
if os.environ.get("MLPG_TIME_VAR"):          # "v_static,v_delta,v_deltadelta": the same for every dimension (the factor's head
    vs = [float(x) for x in os.environ["MLPG_TIME_VAR"].split(",")]          # then has a known length)
    var = torch.tensor(vs, dtype=torch.float64, device=dev).repeat_interleave(dim)
import time                                                     # noqa: E402
t_warm = time.time()                                            # clocks (shader and memory) settle under load: 0.4 s of it
while time.time() - t_warm < float(os.environ.get("MLPG_TIME_WARM_S", "0.4")):
    for _ in range(5):
        ops.mlpg_generation(feat, var, dim, off)
    torch.cuda.synchronize()
times = []
for _ in range(passes):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.mlpg_generation(feat, var, dim, off)
    e1.record()
    e1.synchronize()
    times.append(e0.elapsed_time(e1) * 1e3)
times.sort()
# calls queued back to back (the host side of a call hidden behind the call before it), and what the host side takes
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
n_queue = 16
e0.record()
t_host = time.perf_counter()
for _ in range(n_queue):
    ops.mlpg_generation(feat, var, dim, off)
t_host = (time.perf_counter() - t_host) / n_queue * 1e6
e1.record()
e1.synchronize()
queued = e0.elapsed_time(e1) * 1e3 / n_queue
frames = int(off[-1])
alg = frames * (3 * dim * (4 if f32 else 8) + dim * 8)
print(("float32 rows  " if f32 else "") + ("dim %d  " % dim) + "utterances %d  frames %d  median %.1f us  min %.1f us  p90 %.1f us  algorithmic %.3f GB -> %.2f TB/s at the median"
      % (n_utts, frames, times[len(times) // 2], times[0], times[int(len(times) * 0.9)], alg / 1e9, alg / times[len(times) // 2] / 1e6)
      + "  | %d calls queued: %.1f us a call (%.2f TB/s), host side %.1f us a call" % (n_queue, queued, alg / queued / 1e6, t_host))
