"""Experiment: the per-step LSTM launches are latency-bound -- do two independent half batches on
two streams overlap?  Prints the recurrence (forward and backward) of B rows on one stream, and of
two B/2 halves (interleaved rows, so both halves have the same length profile) on two streams.
usage: python3 scripts/exp_lstm_two_streams.py [T] [B] [H]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from idiaptts_amd import lib as _lib, ops  # noqa: E402
from idiaptts_amd.nn.functional import PackedBatch, _iptr  # noqa: E402


class Job(object):
    def __init__(self, lengths, T, H, dev, L):
        ndir, G = 2, 4
        B = len(lengths)
        self.pb = pb = PackedBatch(lengths, T, False, dev)
        N = pb.N
        self.T, self.B, self.H = T, B, H
        self.gin = torch.randn(N, ndir * G * H, device=dev) * 0.1
        self.whh = torch.randn(ndir, G * H, H, device=dev) * 0.05
        self.y = torch.empty(N, ndir * H, device=dev)
        self.gates = torch.empty(N, ndir * 4 * H, device=dev)
        self.csave = torch.empty(N, ndir * H, device=dev)
        self.dy = torch.randn(N, ndir * H, device=dev)
        self.dg = torch.empty(N, ndir * G * H, device=dev)
        self.state = torch.empty(L.itts_lstm_state_bytes(B, H, ndir), dtype=torch.uint8, device=dev)
        self.L = L

    def fwd(self):
        pb, L = self.pb, self.L
        _lib.check(L.itts_lstm_layer_fwd(_iptr(self.gin), _iptr(self.whh), None, None, _iptr(pb.d_lengths),
                                         pb._hptr(), _iptr(pb.d_row_off), _iptr(pb.d_rev_row), self.T,
                                         self.B, self.H, 2, _iptr(self.y), _iptr(self.gates),
                                         _iptr(self.csave), None, None, _iptr(self.state), ops._stream()), "f")

    def bwd(self):
        pb, L = self.pb, self.L
        _lib.check(L.itts_lstm_layer_bwd(_iptr(self.dy), _iptr(self.whh), None, _iptr(self.gates),
                                         _iptr(self.csave), pb._hptr(), _iptr(pb.d_row_off),
                                         _iptr(pb.d_rev_row), self.T, self.B, self.H, 2, _iptr(self.dg),
                                         None, _iptr(self.state), ops._stream()), "b")


def timeit(fn, n=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return 1e3 * sorted(ts)[len(ts) // 2]


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    H = int(sys.argv[3]) if len(sys.argv) > 3 else 512
    dev = torch.device("cuda:0")
    L = _lib.load()
    lengths = sorted([int(T * (0.3 + 0.7 * i / (B - 1))) for i in range(B)], reverse=True)
    lengths[0] = T
    full = Job(lengths, T, H, dev, L)
    halves = [Job(lengths[0::2], T, H, dev, L), Job(lengths[1::2], lengths[1], H, dev, L)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]

    def two(which):
        cur = torch.cuda.current_stream()
        for j, s in zip(halves, streams):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                getattr(j, which)()
        for s in streams:
            cur.wait_stream(s)

    def serial(which):
        for j in halves:
            getattr(j, which)()

    for which in ("fwd", "bwd"):
        print("%s: B=%d one stream %.2f ms | two halves serial %.2f ms | two halves on two streams %.2f ms" % (
            which, B, timeit(lambda: getattr(full, which)()), timeit(lambda: serial(which)),
            timeit(lambda: two(which))))


main()
