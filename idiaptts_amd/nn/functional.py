"""torch.autograd glue for the HIP acoustic-model kernels (PyTorch-ROCm only provides autograd,
device memory and streams; every forward / backward computation below is a C-ABI call)."""
import ctypes

import torch

from .. import lib as _lib
from .. import ops


class LinearActFunction(torch.autograd.Function):
    """y = act(x W^T + b) on rows (any leading shape); rnn_dyn/FFWrapper.py:63-73."""

    @staticmethod
    def forward(ctx, x, weight, bias, act):
        shape = x.shape
        x2 = x.reshape(-1, shape[-1])
        if x2.stride(-1) != 1:
            x2 = x2.contiguous()
        y = ops.linear_fwd(x2, weight.contiguous(), bias, act)
        ctx.save_for_backward(x2, weight, y)
        ctx.act = act
        ctx.has_bias = bias is not None
        ctx.in_shape = shape
        return y.reshape(*shape[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, weight, y = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1])
        if dy2.stride(-1) != 1:
            dy2 = dy2.contiguous()
        dz = ops.act_bwd(dy2, y, ctx.act) if ctx.act != ops.ACT_NONE else dy2
        dx = dw = db = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw, db = ops.linear_bwd_weight(dz, x2, want_bias=ctx.has_bias)
        if ctx.needs_input_grad[0]:
            dx = ops.linear_bwd_input(dz, weight.contiguous()).reshape(ctx.in_shape)
        return dx, dw, (db if ctx.has_bias else None), None


def _iptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


class LSTMLayerFunction(torch.autograd.Function):
    """One (bi)directional LSTM layer on a time-major padded batch x [T, B, F] with per-row
    lengths (packed-sequence semantics of rnn_dyn/RNNWrapper.py:89-102)."""

    @staticmethod
    def forward(ctx, x, lengths, w_ih, w_hh, b_ih, b_hh, h0, c0, training):
        L = _lib.load()
        T, B, F = x.shape
        ndir, G4, H = w_hh.shape
        x2 = x.contiguous().reshape(T * B, F)
        w_ih_cat = w_ih.reshape(ndir * G4, F)
        gin = ops.linear_fwd(x2, w_ih_cat, (b_ih + b_hh).reshape(-1), ops.ACT_NONE)
        dev = x.device
        y = torch.empty((T * B, ndir * H), dtype=torch.float32, device=dev)
        keep = bool(training)
        gates = torch.empty((T * B, ndir * G4), dtype=torch.float32, device=dev) if keep else None
        csave = torch.empty((T * B, ndir * H), dtype=torch.float32, device=dev) if keep else None
        # zero-filled: padded rows enter the dW_hh GEMM multiplied by zero gate gradients, and
        # 0 * (uninitialised NaN) would poison the sum
        hprev = torch.zeros((T * B, ndir * H), dtype=torch.float32, device=dev) if keep else None
        hn = torch.empty((ndir, B, H), dtype=torch.float32, device=dev)
        cn = torch.empty((ndir, B, H), dtype=torch.float32, device=dev)
        state = torch.empty(L.itts_lstm_state_bytes(B, H, ndir), dtype=torch.uint8, device=dev)
        w_hh_c = w_hh.contiguous()
        h0c = h0.contiguous() if h0 is not None else None
        c0c = c0.contiguous() if c0 is not None else None
        _lib.check(L.itts_lstm_layer_fwd(_iptr(gin), _iptr(w_hh_c), _iptr(h0c), _iptr(c0c),
                                         _iptr(lengths), T, B, H, ndir, _iptr(y), _iptr(gates),
                                         _iptr(csave), _iptr(hprev), _iptr(hn), _iptr(cn),
                                         _iptr(state), ops._stream()), "itts_lstm_layer_fwd")
        if keep:
            ctx.save_for_backward(x2, lengths, w_ih_cat, w_hh_c, gates, csave, hprev,
                                  c0c if c0c is not None else torch.empty(0, device=dev))
            ctx.dims = (T, B, F, H, ndir, c0c is not None)
        return y.reshape(T, B, ndir * H), hn, cn

    @staticmethod
    def backward(ctx, dy, dhn, dcn):
        L = _lib.load()
        x2, lengths, w_ih_cat, w_hh, gates, csave, hprev, c0 = ctx.saved_tensors
        T, B, F, H, ndir, has_c0 = ctx.dims
        G4 = 4 * H
        dev = dy.device
        dy2 = dy.contiguous().reshape(T * B, ndir * H)
        dg = torch.empty((T * B, ndir * G4), dtype=torch.float32, device=dev)
        state = torch.empty(L.itts_lstm_state_bytes(B, H, ndir), dtype=torch.uint8, device=dev)
        w_hh_t = w_hh.transpose(1, 2).contiguous()          # [ndir, H, 4H]
        _lib.check(L.itts_lstm_layer_bwd(_iptr(dy2), _iptr(w_hh_t), _iptr(c0 if has_c0 else None),
                                         _iptr(gates), _iptr(csave), _iptr(lengths), T, B, H, ndir,
                                         _iptr(dg), _iptr(state), ops._stream()),
                   "itts_lstm_layer_bwd")
        dw_ih, db = ops.linear_bwd_weight(dg, x2)                      # [ndir*4H, F], [ndir*4H]
        dw_hh = torch.empty((ndir, G4, H), dtype=torch.float32, device=dev)
        for d in range(ndir):
            ops.linear_bwd_weight(dg[:, d * G4:(d + 1) * G4], hprev[:, d * H:(d + 1) * H],
                                  dw=dw_hh[d], want_bias=False)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops.linear_bwd_input(dg, w_ih_cat).reshape(T, B, F)
        db = db.reshape(ndir, G4)
        return dx, None, dw_ih.reshape(ndir, G4, F), dw_hh, db, db.clone(), None, None, None


class GRULayerFunction(torch.autograd.Function):
    """One (bi)directional GRU layer on a time-major padded batch x [T, B, F] with per-row
    lengths (torch.nn.GRU on packed sequences, rnn_dyn/RNNWrapper.py:45-107)."""

    @staticmethod
    def forward(ctx, x, lengths, w_ih, w_hh, b_ih, b_hh, h0, training):
        L = _lib.load()
        T, B, F = x.shape
        ndir, G3, H = w_hh.shape
        x2 = x.contiguous().reshape(T * B, F)
        w_ih_cat = w_ih.reshape(ndir * G3, F)
        gin = ops.linear_fwd(x2, w_ih_cat, b_ih.reshape(-1), ops.ACT_NONE)
        dev = x.device
        y = torch.empty((T * B, ndir * H), dtype=torch.float32, device=dev)
        keep = bool(training)
        gates = torch.empty((T * B, ndir * G3), dtype=torch.float32, device=dev) if keep else None
        hnpre = torch.empty((T * B, ndir * H), dtype=torch.float32, device=dev) if keep else None
        # zero-filled: padded rows meet zero gate gradients in the dW_hh GEMM (0 * NaN guard)
        hprev = torch.zeros((T * B, ndir * H), dtype=torch.float32, device=dev) if keep else None
        hn = torch.empty((ndir, B, H), dtype=torch.float32, device=dev)
        state = torch.empty(L.itts_gru_state_bytes(B, H, ndir), dtype=torch.uint8, device=dev)
        w_hh_c, b_hh_c = w_hh.contiguous(), b_hh.contiguous()
        h0c = h0.contiguous() if h0 is not None else None
        _lib.check(L.itts_gru_layer_fwd(_iptr(gin), _iptr(w_hh_c), _iptr(b_hh_c), _iptr(h0c),
                                        _iptr(lengths), T, B, H, ndir, _iptr(y), _iptr(gates),
                                        _iptr(hnpre), _iptr(hprev), _iptr(hn), _iptr(state),
                                        ops._stream()), "itts_gru_layer_fwd")
        if keep:
            ctx.save_for_backward(x2, lengths, w_ih_cat, w_hh_c, gates, hnpre, hprev)
            ctx.dims = (T, B, F, H, ndir)
        return y.reshape(T, B, ndir * H), hn

    @staticmethod
    def backward(ctx, dy, dhn):
        L = _lib.load()
        x2, lengths, w_ih_cat, w_hh, gates, hnpre, hprev = ctx.saved_tensors
        T, B, F, H, ndir = ctx.dims
        G3 = 3 * H
        dev = dy.device
        dy2 = dy.contiguous().reshape(T * B, ndir * H)
        dgi = torch.empty((T * B, ndir * G3), dtype=torch.float32, device=dev)
        dgh = torch.empty((T * B, ndir * G3), dtype=torch.float32, device=dev)
        state = torch.empty(L.itts_gru_state_bytes(B, H, ndir), dtype=torch.uint8, device=dev)
        w_hh_t = w_hh.transpose(1, 2).contiguous()          # [ndir, H, 3H]
        _lib.check(L.itts_gru_layer_bwd(_iptr(dy2), _iptr(w_hh_t), _iptr(gates), _iptr(hnpre),
                                        _iptr(hprev), _iptr(lengths), T, B, H, ndir, _iptr(dgi),
                                        _iptr(dgh), _iptr(state), ops._stream()),
                   "itts_gru_layer_bwd")
        dw_ih, db_ih = ops.linear_bwd_weight(dgi, x2)                  # [ndir*3H, F], [ndir*3H]
        dw_hh = torch.empty((ndir, G3, H), dtype=torch.float32, device=dev)
        db_hh = torch.empty((ndir, G3), dtype=torch.float32, device=dev)
        for d in range(ndir):
            _, db = ops.linear_bwd_weight(dgh[:, d * G3:(d + 1) * G3], hprev[:, d * H:(d + 1) * H],
                                          dw=dw_hh[d])
            db_hh[d] = db
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops.linear_bwd_input(dgi, w_ih_cat).reshape(T, B, F)
        return dx, None, dw_ih.reshape(ndir, G3, F), dw_hh, db_ih.reshape(ndir, G3), db_hh, \
            None, None
