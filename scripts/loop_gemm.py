import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from idiaptts_amd import ops
dev = torch.device("cuda:0")
M = 39129
h1 = torch.tanh(torch.randn(M, 512, device=dev)); w2 = torch.randn(512, 512, device=dev) * 0.05; b = torch.zeros(512, device=dev)
o = torch.empty(M, 512, device=dev)
t0 = time.time()
n = 0
while time.time() - t0 < float(sys.argv[1]):
    for _ in range(200): ops.linear_fwd(h1, w2, b, 1, out=o)
    torch.cuda.synchronize(); n += 200
print("launches", n, "avg us", (time.time() - t0) / n * 1e6)
