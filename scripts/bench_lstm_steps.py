"""Times the recurrence of one BiLSTM / BiGRU layer (no GEMMs): T steps, B rows, H units, equal
lengths (every tile active).  usage: python3 scripts/bench_lstm_steps.py [T] [B] [H]"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from idiaptts_amd import lib as _lib, ops  # noqa: E402
from idiaptts_amd.nn.functional import PackedBatch, _iptr  # noqa: E402


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    H = int(sys.argv[3]) if len(sys.argv) > 3 else 512
    dev = torch.device("cuda:0")
    L = _lib.load()
    ndir = 2
    pb = PackedBatch([T] * B, T, False, dev)
    N = pb.N
    for cell, G in (("lstm", 4), ("gru", 3)):
        gin = torch.randn(N, ndir * G * H, device=dev) * 0.1
        whh = torch.randn(ndir, G * H, H, device=dev) * 0.05
        bhh = torch.zeros(ndir, G * H, device=dev)
        y = torch.empty(N, ndir * H, device=dev)
        gates = torch.empty(N, ndir * 4 * H, device=dev)
        aux1 = torch.empty(N, ndir * H, device=dev)
        aux2 = torch.empty(N, ndir * H, device=dev)
        dy = torch.randn(N, ndir * H, device=dev)
        dg = torch.empty(N, ndir * G * H, device=dev)
        dg2 = torch.empty(N, ndir * G * H, device=dev)
        nbytes = max(L.itts_lstm_state_bytes(B, H, ndir), L.itts_gru_state_bytes(B, H, ndir))
        state = torch.empty(nbytes, dtype=torch.uint8, device=dev)

        def fwd(train):
            g, a1 = (gates, aux1) if train else (None, None)
            if cell == "lstm":
                _lib.check(L.itts_lstm_layer_fwd(_iptr(gin), _iptr(whh), None, None,
                                                 _iptr(pb.d_lengths), pb._hptr(), _iptr(pb.d_row_off),
                                                 _iptr(pb.d_rev_row), T, B, H, ndir, _iptr(y), _iptr(g),
                                                 _iptr(a1), None, None, _iptr(state),
                                                 ops._stream()), "f")
            else:
                _lib.check(L.itts_gru_layer_fwd(_iptr(gin), _iptr(whh), _iptr(bhh), None,
                                                _iptr(pb.d_lengths), pb._hptr(), _iptr(pb.d_row_off),
                                                _iptr(pb.d_rev_row), T, B, H, ndir, _iptr(y), _iptr(g),
                                                None, _iptr(state), ops._stream()), "f")

        def bwd():
            if cell == "lstm":
                _lib.check(L.itts_lstm_layer_bwd(_iptr(dy), _iptr(whh), None, _iptr(gates),
                                                 _iptr(aux1), pb._hptr(), _iptr(pb.d_row_off),
                                                 _iptr(pb.d_rev_row), T, B, H, ndir, _iptr(dg),
                                                 None, _iptr(state), ops._stream()), "b")
            else:
                _lib.check(L.itts_gru_layer_bwd(_iptr(dy), _iptr(whh), _iptr(gates),
                                                _iptr(aux2), pb._hptr(), _iptr(pb.d_row_off),
                                                _iptr(pb.d_rev_row), T, B, H, ndir, _iptr(dg),
                                                _iptr(dg2), None, _iptr(state), ops._stream()), "b")

        for name, fn in (("fwd(train)", lambda: fwd(True)), ("fwd(infer)", lambda: fwd(False)),
                         ("bwd", bwd)):
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 3
            print("%s %-11s T=%d B=%d H=%d: %.2f us/step" % (cell, name, T, B, H, dt / T * 1e6))


if __name__ == "__main__":
    main()
