"""Experiment: does a weight-gradient GEMM on a side stream slow the (latency-bound) LSTM backward
recurrence on the main stream?  Prints recurrence alone, GEMMs alone, both serial, both overlapped.
usage: python3 scripts/exp_lstm_overlap.py [T] [B] [H]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from idiaptts_amd import lib as _lib, ops  # noqa: E402
from idiaptts_amd.nn.functional import PackedBatch, _iptr  # noqa: E402


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    H = int(sys.argv[3]) if len(sys.argv) > 3 else 512
    dev = torch.device("cuda:0")
    L = _lib.load()
    ndir, G = 2, 4
    lengths = sorted([int(T * (0.3 + 0.7 * i / (B - 1))) for i in range(B)], reverse=True)
    lengths[0] = T
    pb = PackedBatch(lengths, T, False, dev)
    N = pb.N
    gin = torch.randn(N, ndir * G * H, device=dev) * 0.1
    whh = torch.randn(ndir, G * H, H, device=dev) * 0.05
    y = torch.empty(N, ndir * H, device=dev)
    gates = torch.empty(N, ndir * 4 * H, device=dev)
    csave = torch.empty(N, ndir * H, device=dev)
    dy = torch.randn(N, ndir * H, device=dev)
    dg = torch.empty(N, ndir * G * H, device=dev)
    state = torch.empty(L.itts_lstm_state_bytes(B, H, ndir), dtype=torch.uint8, device=dev)
    _lib.check(L.itts_lstm_layer_fwd(_iptr(gin), _iptr(whh), None, None, _iptr(pb.d_lengths), pb._hptr(),
                                     _iptr(pb.d_row_off), _iptr(pb.d_rev_row), T, B, H, ndir, _iptr(y),
                                     _iptr(gates), _iptr(csave), None, None, _iptr(state), ops._stream()), "f")
    # the dW GEMMs of the layer above: dG^T [8H, N] x X [N, 2H]
    dg_up = torch.randn(N, ndir * G * H, device=dev)
    x_up = torch.randn(N, 2 * H, device=dev)
    dw = torch.zeros(ndir * G * H, 2 * H, device=dev)
    db = torch.zeros(ndir * G * H, device=dev)
    side = torch.cuda.Stream()

    def rec():
        _lib.check(L.itts_lstm_layer_bwd(_iptr(dy), _iptr(whh), None, _iptr(gates), _iptr(csave), pb._hptr(),
                                         _iptr(pb.d_row_off), _iptr(pb.d_rev_row), T, B, H, ndir, _iptr(dg),
                                         None, _iptr(state), ops._stream()), "b")

    def gemm():
        ops.linear_bwd_weight(dg_up, x_up, dw=dw, db=db)

    def serial():
        gemm()
        rec()

    def overlapped():
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            gemm()
        rec()
        torch.cuda.current_stream().wait_stream(side)

    for name, fn in (("recurrence", rec), ("dW gemm", gemm), ("serial", serial), ("overlapped", overlapped)):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        print("%-11s %8.2f ms  (N=%d rows, T=%d)" % (name, (time.perf_counter() - t0) / 3 * 1e3, N, T))


if __name__ == "__main__":
    main()
