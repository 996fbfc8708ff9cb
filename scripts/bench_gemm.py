import os, sys, torch, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from idiaptts_amd import ops
dev = torch.device("cuda:0")
M = 39129
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / n
g = torch.Generator(device="cpu").manual_seed(0)
def R(*s): return (torch.rand(*s, generator=g) - 0.5).to(dev)
def P(t):   # zero-padded pitch (multiple of 4 floats), as FlatFFModel lays its buffers out
    p = torch.zeros(t.shape[0], (t.shape[1] + 3) // 4 * 4, device=dev); p[:, :t.shape[1]] = t
    return p[:, :t.shape[1]]
x = P(R(M, 425)); h1 = torch.tanh(R(M, 512)); h2 = torch.tanh(R(M, 512)); dz3 = P(R(M, 187)); dz2 = R(M, 512); dz1 = R(M, 512)
w1 = P(R(512, 425)); w2 = R(512, 512); w3 = R(187, 512); b1 = R(512); b3 = R(187)
w1 = torch.zeros(512, 428, device=dev); x = torch.zeros(M, 428, device=dev); x[:, :425] = R(M, 425); w1[:, :425] = R(512, 425)
o512 = torch.empty(M, 512, device=dev); o187 = P(torch.zeros(M, 187, device=dev))
dw = {k: torch.empty(v.shape, device=dev) for k, v in dict(w1=w1, w2=w2, w3=w3).items()}
cases = [
 ("fwd1 M,512,K425", lambda: ops.linear_fwd(x, w1, b1, 1, out=o512), 2*M*512*428),
 ("fwd2 M,512,K512", lambda: ops.linear_fwd(h1, w2, b1, 1, out=o512), 2*M*512*512),
 ("fwd3 M,187,K512", lambda: ops.linear_fwd(h2, w3, b3, 0, out=o187), 2*M*187*512),
 ("dW3 187x512", lambda: ops.linear_bwd_weight(dz3, h2, dw=dw["w3"], want_bias=False), 2*M*187*512),
 ("dX3 M,512,red187", lambda: ops.linear_bwd_input(dz3, w3, yprev=h2, act_prev=1, out=o512), 2*M*187*512),
 ("dW2 512x512", lambda: ops.linear_bwd_weight(dz2, h1, dw=dw["w2"], want_bias=False), 2*M*512*512),
 ("dX2 M,512,red512", lambda: ops.linear_bwd_input(dz2, w2, yprev=h1, act_prev=1, out=o512), 2*M*512*512),
 ("dW1 512x425", lambda: ops.linear_bwd_weight(dz1, x, dw=dw["w1"], want_bias=False), 2*M*512*428),
]
tot = 0
for name, fn, fl in cases:
    ms = t(fn); tot += ms
    print("%-22s %8.1f us  %6.1f TF/s" % (name, ms*1e3, fl/ms/1e9))
print("total %.1f us" % (tot*1e3))
