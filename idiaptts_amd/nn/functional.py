"""torch.autograd glue for the HIP acoustic-model kernels (PyTorch-ROCm only provides autograd,
device memory and streams; every forward / backward computation below is a C-ABI call)."""
import collections
import os
import ctypes
import threading

import torch

from .. import lib as _lib
from .. import ops


# The gradient a scalar loss's backward starts from, as an object the loss functions can recognise: `loss.backward()`
# makes a fresh tensor of ones per call, and a Function's backward cannot tell ones from any other factor without
# reading the value back, so it multiplies its stored gradient by it -- a pass over [frames x features] for a factor
# of 1.  The training loop starts backward from `unit_gradient(loss)` instead; autograd hands that very tensor to
# the root's backward, which then skips the product (x * 1.0 is x).
_unit_gradients = {}


def unit_gradient(like):
    key = (like.device, like.dtype)
    unit = _unit_gradients.get(key)
    if unit is None:
        unit = _unit_gradients[key] = torch.ones((), dtype=like.dtype, device=like.device)
    return unit


def is_unit_gradient(grad):
    unit = _unit_gradients.get((grad.device, grad.dtype))
    return unit is not None and grad.dim() == 0 and grad.data_ptr() == unit.data_ptr()


class LinearActFunction(torch.autograd.Function):
    """y = act(x W^T + b) on rows (any leading shape); rnn_dyn/FFWrapper.py:63-73.
    An input whose last extent is `in_features` rounded up to a multiple of four (and not `in_features` itself:
    425 -> 428) is taken as rows with ZEROED pad columns (ValidRows.pack writes them): the weight gets the same zero
    columns for the products, so every operand has a 16-byte row pitch and the GEMM entry points take their LDS-DMA
    kernels -- the register-staged ones took 201 instead of 110 us for the first layer of the 425-512-512-187 model
    and 179 instead of 125 for its weight gradient.  The zeros add nothing to any sum: the same bits."""

    @staticmethod
    def forward(ctx, x, weight, bias, act):
        shape = x.shape
        x2 = x.reshape(-1, shape[-1])
        if x2.stride(-1) != 1:
            x2 = x2.contiguous()
        N, K = weight.shape
        w = weight.contiguous()
        ctx.k_pad = 0
        if shape[-1] != K and shape[-1] == (K + 3) // 4 * 4:
            ctx.k_pad = shape[-1] - K
            w = torch.nn.functional.pad(w, (0, ctx.k_pad))
        # an output width that is no multiple of four floats (187: the acoustic features) gets rows of a
        # 16-byte multiple (the result is a view of them): the GEMM entry points then take their LDS-DMA
        # kernel instead of the register-staged one -- 0.7 + 0.6 ms per BiLSTM training step otherwise
        out = None
        if N % 4:
            full = torch.empty((x2.shape[0], (N + 3) // 4 * 4), dtype=torch.float32, device=x2.device)
            full[:, N:] = 0          # pad columns: read again by act_bwd over the whole rows
            out = full[:, :N]
        y = ops.linear_fwd(x2, w, bias, act, out=out)
        ctx.save_for_backward(x2, w, y)
        ctx.act = act
        ctx.has_bias = bias is not None
        ctx.in_shape = shape
        # (callers get contiguous rows, as from torch's own linear: a `.view` on the result must keep working)
        return (y if out is None else y.contiguous()).reshape(*shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, w, y = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1])
        if dy2.stride(-1) != 1:
            dy2 = dy2.contiguous()
        if dy2.stride(0) % 4:                      # the same for the incoming gradient (pad columns zero)
            N = dy2.shape[1]
            buf = torch.empty((dy2.shape[0], (N + 3) // 4 * 4), dtype=torch.float32, device=dy2.device)
            buf[:, N:] = 0
            buf[:, :N] = dy2
            dy2 = buf[:, :N]
        dz = ops.act_bwd(dy2, y, ctx.act) if ctx.act != ops.ACT_NONE else dy2
        dx = dw = db = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw, db = ops.linear_bwd_weight(dz, x2, want_bias=ctx.has_bias)
            if ctx.k_pad:
                dw = dw[:, :dw.shape[1] - ctx.k_pad]
        if ctx.needs_input_grad[0]:
            dx = ops.linear_bwd_input(dz, w).reshape(ctx.in_shape)
        return dx, dw, (db if ctx.has_bias else None), None


def _rows_padded(M, N, device):
    """[M, N] view of a fresh buffer whose row pitch is N rounded up to four floats, pad columns zero"""
    Np = (N + 3) // 4 * 4
    full = torch.empty((M, Np), dtype=torch.float32, device=device)
    if Np != N:
        full[:, N:] = 0
    return full[:, :N]


class LinearChainFunction(torch.autograd.Function):
    """A run of Linear (+ Tanh / ReLU) layers on rows -- the layers of consecutive FFWrapper groups
    (rnn_dyn/FFWrapper.py:63-73) -- as ONE autograd node: forward the same `itts_linear_fwd` launches as the layers
    one by one (the same bits), backward as the flat step runs it (native_ff.py): per layer ONE launch for weight
    gradient, bias gradient and input gradient (`itts_linear_bwd`), the previous layer's activation derivative in its
    epilogue -- no `act_bwd` pass over the rows, no second launch per layer, one node instead of one per layer.
    x: [M, K0] rows, or [M, K0 rounded up to 4] with zeroed pad columns (LinearActFunction's convention);
    acts: tuple of ops.ACT_*; wb: weight0, bias0, weight1, bias1, ..."""

    @staticmethod
    def forward(ctx, x, acts, *wb):
        n = len(acts)
        if x.stride(-1) != 1:
            x = x.contiguous()
        ws, hs, h = [], [], x
        for i in range(n):
            w, b = wb[2 * i], wb[2 * i + 1]
            N, K = w.shape
            w = w.contiguous()
            if i == 0 and h.shape[1] != K and h.shape[1] == (K + 3) // 4 * 4:
                w = torch.nn.functional.pad(w, (0, h.shape[1] - K))
            h = ops.linear_fwd(h, w, b, acts[i], out=_rows_padded(h.shape[0], N, h.device))
            ws.append(w)
            hs.append(h)
        ctx.save_for_backward(x, *hs, *ws)
        ctx.acts, ctx.n = tuple(acts), n
        ctx.k0 = wb[0].shape[1]
        return hs[-1]

    @staticmethod
    def backward(ctx, dy):
        n = ctx.n
        saved = ctx.saved_tensors
        x, hs, ws = saved[0], saved[1:1 + n], saved[1 + n:]
        M = x.shape[0]
        dz = dy
        if dz.stride(-1) != 1 or dz.stride(0) % 4:
            buf = _rows_padded(M, dy.shape[1], dy.device)
            buf.copy_(dy)
            dz = buf
        if ctx.acts[-1] != ops.ACT_NONE:
            dz = ops.act_bwd(dz, hs[-1], ctx.acts[-1])
        grads = [None] * (2 * n)
        dx = None
        for i in range(n - 1, -1, -1):
            w = ws[i]
            N, K = w.shape
            dw = torch.empty((N, K), dtype=torch.float32, device=x.device)
            db = torch.empty((N,), dtype=torch.float32, device=x.device)
            if i > 0:
                dz_in = _rows_padded(M, K, x.device)
                ops.linear_bwd(dz, hs[i - 1], w, dw, db, dz_in, yprev=hs[i - 1], act_prev=ctx.acts[i - 1])
                grads[2 * i], grads[2 * i + 1] = dw, db
                dz = dz_in
            else:
                if ctx.needs_input_grad[0]:
                    dx = torch.empty((M, K), dtype=torch.float32, device=x.device)
                    ops.linear_bwd(dz, x, w, dw, db, dx)
                else:
                    ops.linear_bwd_weight(dz, x, dw=dw, db=db)
                grads[0], grads[1] = (dw[:, :ctx.k0] if K != ctx.k0 else dw), db
        return (dx, None) + tuple(grads)


def _iptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


class PackedBatch(object):
    """Index bookkeeping of pack_padded_sequence(enforce_sorted=False) (rnn_dyn/RNNWrapper.py:89-92)
    for a padded batch: rows sorted by decreasing length (stable), packed row of frame t of sorted
    row b = row_off[t] + b.  `flat_index` maps every packed row to its row in the padded tensor
    flattened over (time, batch) -- or (batch, time) when batch_first -- so packing is one
    index_select and unpacking one index_copy."""

    _cache = collections.OrderedDict()      # (lengths, padded_time, batch_first, device) -> instance
    _cache_max = 16

    @classmethod
    def get(cls, lengths, padded_time, batch_first, device):
        """The bookkeeping of a batch whose lengths were seen recently is reused (its tables are ~1 ms of
        numpy and eight small uploads during which the GPU has nothing to do: validation passes and
        trainers that keep their batches fixed over the epochs meet the same length vectors again)."""
        import numpy as np
        lens = np.asarray(torch.as_tensor(lengths).cpu(), dtype=np.int64)
        key = (lens.tobytes(), int(padded_time), bool(batch_first), str(device))
        pb = cls._cache.get(key)
        if pb is None:
            pb = cls(lens, padded_time, batch_first, device)
            cls._cache[key] = pb
            while len(cls._cache) > cls._cache_max:
                cls._cache.popitem(last=False)
        else:
            cls._cache.move_to_end(key)
        return pb

    def __init__(self, lengths, padded_time, batch_first, device):
        import numpy as np
        lengths = np.asarray(torch.as_tensor(lengths).cpu(), dtype=np.int64)
        if lengths.ndim != 1 or len(lengths) == 0 or lengths.min() < 1:
            raise ValueError("lengths must be a non-empty vector of positive values")
        if lengths.max() > padded_time:
            raise ValueError("a length exceeds the padded time extent")
        self.B, self.T = len(lengths), int(lengths.max())
        perm = np.argsort(-lengths, kind="stable")
        sorted_len = lengths[perm]
        nact = (sorted_len[None, :] > np.arange(self.T)[:, None]).sum(axis=1)      # [T]
        row_off = np.concatenate([[0], np.cumsum(nact)[:-1]])
        self.N = int(nact.sum())
        t_of = np.repeat(np.arange(self.T), nact)
        b_of = np.arange(self.N) - row_off[t_of]
        flat = perm[b_of] * padded_time + t_of if batch_first else t_of * self.B + perm[b_of]
        self.h_lengths = torch.from_numpy(sorted_len.astype(np.int32))            # host, sorted
        self.d_lengths = self.h_lengths.to(device)
        self.d_row_off = torch.from_numpy(row_off.astype(np.int32)).to(device)
        # packed row the reverse direction visits at step s for sorted row b (0 where inactive)
        t_rev = np.maximum(sorted_len[None, :] - 1 - np.arange(self.T)[:, None], 0)  # [T, B]
        rev = row_off[t_rev] + np.arange(self.B)[None, :]
        self.d_rev_row = torch.from_numpy(np.ascontiguousarray(rev, dtype=np.int32)).to(device)
        self.flat_index = torch.from_numpy(flat.astype(np.int64)).to(device)
        # padded position -> packed row (-1: padding), for the way back
        inv = np.full(self.B * int(padded_time), -1, dtype=np.int64)
        inv[flat] = np.arange(self.N)
        self.inv_flat = torch.from_numpy(inv).to(device)
        # packed row of the previous frame of the same sequence in each direction's visiting order
        # (forward: t - 1, reverse: t + 1); first frames point at row N, where shift() puts h0
        prev_f = np.where(t_of > 0, row_off[np.maximum(t_of - 1, 0)] + b_of, self.N)
        nxt = np.minimum(t_of + 1, self.T - 1)
        prev_r = np.where(t_of + 1 < sorted_len[b_of], row_off[nxt] + b_of, self.N)
        self.prev_row = [torch.from_numpy(p_.astype(np.int64)).to(device) for p_ in (prev_f, prev_r)]
        self.perm = torch.from_numpy(perm.astype(np.int64)).to(device)
        self.inv_perm = torch.from_numpy(np.argsort(perm).astype(np.int64)).to(device)

    def pack(self, padded, pad_cols=False):
        """[T, B, F] (or [B, T, F]) -> [N, F]; pad_cols: -> [N, F rounded up to a multiple of 4] with zero
        columns (rows of 16-byte multiples let the GEMM entry points take their LDS-DMA kernel)."""
        flat2 = padded.reshape(-1, padded.shape[-1])
        F = flat2.shape[1]
        return RowsGatherFunction.apply(flat2, self.flat_index, self.inv_flat, (F + 3) // 4 * 4 if pad_cols else F)

    def unpack(self, packed, padded_shape):
        """[N, D] -> zero-padded [T, B, D] / [B, T, D] (pad_packed_sequence)"""
        out = RowsGatherFunction.apply(packed, self.inv_flat, self.flat_index, packed.shape[-1])
        return out.reshape(padded_shape[0], padded_shape[1], packed.shape[-1])

    def shift(self, y, h0, ndir, H):
        """h_{t-1} of every packed frame: the layer output [N, ndir*H] moved by one frame along each
        sequence (per direction), the initial state h0 [ndir, H] (or zeros) at the first frame."""
        out = torch.empty_like(y)
        for d in range(ndir):
            ops.rows_gather(y[:, d * H:(d + 1) * H], self.prev_row[d], fill_row=h0[d] if h0 is not None else None,
                            out=out[:, d * H:(d + 1) * H])
        return out

    def first_rows(self, d):
        """Packed rows of the first frame direction d processes in every sequence ([B] indices):
        where the recurrence starts from the initial state."""
        cache = self.__dict__.setdefault("_first_rows", {})
        if d not in cache:
            cache[d] = (self.prev_row[d] == self.N).nonzero().reshape(-1)
        return cache[d]

    def _hptr(self):
        return ctypes.c_void_p(self.h_lengths.data_ptr())


_padding_state = threading.local()


class padding_rows_identical(object):
    """Context in which the caller vouches that all padding positions of the padded batches it hands to the model
    hold the same row (true of ModularModelHandlerPyTorch.prepare_batch: pad_sequence writes zeros): the
    frame-independent layer groups then run on the valid rows plus one representative row (ValidRows) instead of
    every position of the padded tensor.  Outside such a context they compute all positions, as the reference
    does."""

    def __init__(self, on=True):
        self.on = bool(on)

    def __enter__(self):
        self.prev = getattr(_padding_state, "identical", False)
        _padding_state.identical = self.on
        return self

    def __exit__(self, *exc):
        _padding_state.identical = self.prev


def padding_is_identical():
    return getattr(_padding_state, "identical", False)


class ValidRows(object):
    """Bookkeeping of a padded batch for layers that treat every frame on its own (the Linear groups of
    rnn_dyn/FFWrapper.py): they run on the VALID rows of the batch, stored back to back in the batch's own order
    (utterance b at rows starts[b] ..), plus ONE row for all the padding positions -- the reference computes every
    padding position of [B, T, F] (a third of an LJSpeech batch of 32 utterances), and every one of them holds the
    same row (pad_sequence / pad_packed_sequence write zeros, an embedding of a padded index the same vector), so
    one representative gives their common value: `representative` is the flat position (in the batch's layout) of
    the last frame of the shortest utterance.  Unlike PackedBatch nothing is sorted and nothing but the lengths is
    uploaded: starts / lens are what the kernels of csrc/batch_rows.hip index with."""

    _cache = collections.OrderedDict()
    _cache_max = 16

    @classmethod
    def get(cls, lengths, padded_time, batch_first, device):
        import numpy as np
        lens = np.asarray(torch.as_tensor(lengths).cpu(), dtype=np.int64)
        key = (lens.tobytes(), int(padded_time), bool(batch_first), str(device))
        vr = cls._cache.get(key)
        if vr is None:
            vr = cls(lens, padded_time, batch_first, device)
            cls._cache[key] = vr
            while len(cls._cache) > cls._cache_max:
                cls._cache.popitem(last=False)
        else:
            cls._cache.move_to_end(key)
        return vr

    def __init__(self, lengths, padded_time, batch_first, device):
        import numpy as np
        if lengths.ndim != 1 or len(lengths) == 0 or lengths.min() < 0:
            raise ValueError("lengths must be a non-empty vector of non-negative values")
        if lengths.max() > padded_time:
            raise ValueError("a length exceeds the padded time extent")
        self.B, self.T, self.batch_first = len(lengths), int(padded_time), bool(batch_first)
        self.N = int(lengths.sum())
        self.n_pad = self.B * self.T - self.N
        table = np.empty((2, self.B), dtype=np.int64)
        table[0, 0] = 0
        np.cumsum(lengths[:-1], out=table[0, 1:])
        table[1] = lengths
        host = torch.from_numpy(table)
        if torch.cuda.is_available():
            host = host.pin_memory()
        self.table = host.to(device, non_blocking=True)       # ONE upload: starts, lens
        self.starts, self.lens = self.table[0], self.table[1]
        b = int(np.argmin(lengths))
        self.representative = (b * self.T + self.T - 1) if batch_first else ((self.T - 1) * self.B + b)

    def pack(self, padded, pad_cols=True):
        """[B, T, F] / [T, B, F] -> [N (+ 1 when the batch has padding), F rounded up to a multiple of 4]"""
        return PackValidFunction.apply(padded, self, pad_cols)

    def unpack(self, rows):
        """[N (+ 1), D] -> [B, T, D] / [T, B, D]; the padding positions take the last row's value"""
        return UnpackValidFunction.apply(rows, self)


class PackValidFunction(torch.autograd.Function):
    """Valid rows of a padded batch back to back, then (when the batch has padding) the representative padding
    row.  Backward: valid positions take their row's gradient; the gradient of the extra row -- the SUM over all
    padding positions of what the layers passed back -- lands on the representative position, the other padding
    positions get zero: whatever produced the identical padding rows sees the same total."""

    @staticmethod
    def forward(ctx, padded, vr, pad_cols):
        if not padded.is_contiguous():
            padded = padded.contiguous()
        F = padded.shape[2]
        width = (F + 3) // 4 * 4 if pad_cols else F
        out = torch.empty((vr.N + (1 if vr.n_pad else 0), width), dtype=torch.float32, device=padded.device)
        ops.batch_pack_rows(padded, vr.starts, vr.lens, vr.batch_first, vr.N, out_width=width, out=out,
                            rep_pos=vr.representative if vr.n_pad else -1, rep_dst_row=vr.N)
        ctx.vr, ctx.F = vr, F
        return out             # (with its zeroed pad columns: LinearActFunction pads its weight to match)

    @staticmethod
    def backward(ctx, grad):
        vr, F = ctx.vr, ctx.F
        g = grad[:, :F] if grad.stride(-1) == 1 else grad[:, :F].contiguous()
        dx, _ = ops.batch_pad_gather(g[:vr.N], vr.starts, vr.lens, vr.B, vr.T, vr.batch_first, width=F,
                                     rep_pos=vr.representative if vr.n_pad else -1,
                                     rep_row=g[vr.N] if vr.n_pad else None)
        return dx, None, None


class UnpackValidFunction(torch.autograd.Function):
    """[N (+ 1), D] rows -> the padded batch, every padding position holding the extra row (zeros without one).
    Backward: the rows of the valid positions, and for the extra row the column sums over the padding
    positions (csrc/batch_rows.hip: fixed summation order)."""

    @staticmethod
    def forward(ctx, rows, vr):
        if rows.stride(-1) != 1:
            rows = rows.contiguous()
        fill = rows[vr.N] if vr.n_pad else None
        out, _ = ops.batch_pad_gather(rows[:vr.N], vr.starts, vr.lens, vr.B, vr.T, vr.batch_first, fill_row=fill)
        ctx.vr = vr
        return out

    @staticmethod
    def backward(ctx, grad):
        vr = ctx.vr
        if not grad.is_contiguous():
            grad = grad.contiguous()
        D = grad.shape[2]
        # rows of 16-byte multiples for the GEMMs of the layers' backward (LinearActFunction.backward pads otherwise)
        width = (D + 3) // 4 * 4
        full = torch.empty((vr.N + (1 if vr.n_pad else 0), width), dtype=torch.float32, device=grad.device)
        ops.batch_pack_rows(grad, vr.starts, vr.lens, vr.batch_first, vr.N, out_width=width, out=full)
        if vr.n_pad:
            ops.batch_pad_colsum(grad, vr.lens, vr.batch_first, out=full[vr.N])
        return full[:, :D], None


class RowsGatherFunction(torch.autograd.Function):
    """out[r] = src[idx[r]] (zeros where idx[r] < 0), optionally widened by zero columns; the gradient
    is the gather through the inverse map `idx_back` (pack and unpack are each other's adjoint)."""

    @staticmethod
    def forward(ctx, src, idx, idx_back, out_width):
        ctx.save_for_backward(idx_back)
        ctx.width = src.shape[1]
        if src.stride(-1) != 1:
            src = src.contiguous()
        return ops.rows_gather(src, idx, out_width=out_width)

    @staticmethod
    def backward(ctx, grad):
        (idx_back,) = ctx.saved_tensors
        g = grad if grad.stride(-1) == 1 else grad.contiguous()
        return ops.rows_gather(g, idx_back, width=ctx.width), None, None, None


class StatesToCallerOrder(torch.autograd.Function):
    """The final states of the layers -- each [ndir, B, Hl] in the batch's sorted row order, Hl >= H when the
    hidden size was padded -- as ONE [layers * ndir, B, H] tensor in the caller's row order (what torch.nn.LSTM /
    GRU return as h_n / c_n): one native row gather per layer straight into its slice (the index_select per
    layer and the torch.cat over the layers were the last torch kernels between the recurrent layers)."""

    @staticmethod
    def forward(ctx, inv_perm, perm, H, *states):
        ndir, B = states[0].shape[0], states[0].shape[1]
        out = torch.empty((len(states) * ndir, B, H), dtype=torch.float32, device=states[0].device)
        o2 = out.view(-1, H)
        for i, st in enumerate(states):
            st = st if st.is_contiguous() else st.contiguous()
            for d in range(ndir):
                ops.rows_gather(st[d], inv_perm, width=H, out=o2[(i * ndir + d) * B:(i * ndir + d + 1) * B])
        ctx.save_for_backward(perm)
        ctx.meta = (ndir, B, H, [st.shape[2] for st in states])
        return out

    @staticmethod
    def backward(ctx, grad):
        (perm,) = ctx.saved_tensors
        ndir, B, H, widths = ctx.meta
        g2 = (grad if grad.is_contiguous() else grad.contiguous()).view(-1, H)
        outs = []
        for i, Hl in enumerate(widths):
            g = torch.empty((ndir, B, Hl), dtype=torch.float32, device=grad.device)
            for d in range(ndir):
                ops.rows_gather(g2[(i * ndir + d) * B:(i * ndir + d + 1) * B], perm, width=H, out=g[d], out_width=Hl)
            outs.append(g)
        return (None, None, None) + tuple(outs)


def _pad4_cols(t):
    """[R, F] -> [R, F rounded up to a multiple of 4] (zero columns): rows of 16-byte multiples let
    the GEMM entry points take their LDS-DMA kernel; F = 425 (the question labels) otherwise sends
    the first recurrent layer's three GEMMs through the register-staged one (3.2 + 2.7 + 0.7 ms
    instead of ~2.1 + 2.3 + 0.5 ms per training step at the bench size)."""
    F = t.shape[1]
    if F % 4 == 0:
        return t
    return torch.nn.functional.pad(t, (0, 4 - F % 4))


class LSTMLayerFunction(torch.autograd.Function):
    """One (bi)directional LSTM layer on packed rows x [N, F] (see PackedBatch)."""

    @staticmethod
    def forward(ctx, x2, pb, w_ih, w_hh, b_ih, b_hh, h0, c0, training):
        L = _lib.load()
        N = x2.shape[0]
        F = w_ih.shape[-1]               # x2 may arrive with its rows already padded to 16-byte multiples
        ndir, G4, H = w_hh.shape
        T, B = pb.T, pb.B
        pre_padded = x2.shape[1] != F
        x2 = _pad4_cols(x2.contiguous())
        w_ih_cat = _pad4_cols(w_ih.reshape(ndir * G4, F))
        gin = ops.linear_fwd(x2, w_ih_cat, (b_ih + b_hh).reshape(-1), ops.ACT_NONE)
        dev = x2.device
        y = torch.empty((N, ndir * H), dtype=torch.float32, device=dev)
        keep = bool(training)
        gates = torch.empty((N, ndir * G4), dtype=torch.float32, device=dev) if keep else None
        csave = torch.empty((N, ndir * H), dtype=torch.float32, device=dev) if keep else None
        hn = torch.empty((ndir, B, H), dtype=torch.float32, device=dev)
        cn = torch.empty((ndir, B, H), dtype=torch.float32, device=dev)
        state = torch.empty(L.itts_lstm_state_bytes(B, H, ndir), dtype=torch.uint8, device=dev)
        w_hh_c = w_hh.contiguous()
        h0c = h0.contiguous() if h0 is not None else None
        c0c = c0.contiguous() if c0 is not None else None
        _lib.check(L.itts_lstm_layer_fwd(_iptr(gin), _iptr(w_hh_c), _iptr(h0c), _iptr(c0c),
                                         _iptr(pb.d_lengths), pb._hptr(), _iptr(pb.d_row_off),
                                         _iptr(pb.d_rev_row), T, B, H, ndir, _iptr(y),
                                         _iptr(gates), _iptr(csave), _iptr(hn), _iptr(cn),
                                         _iptr(state), ops._stream()), "itts_lstm_layer_fwd")
        ctx.set_materialize_grads(False)      # unused h_n / c_n arrive as None in backward
        if not keep:
            ctx.mark_non_differentiable(y, hn, cn)
        if keep:
            empty = torch.empty(0, device=dev)
            ctx.save_for_backward(x2, w_ih_cat, w_hh_c, gates, csave, y,
                                  h0c if h0c is not None else empty,
                                  c0c if c0c is not None else empty)
            ctx.pb = pb
            ctx.dims = (F, H, ndir, h0c is not None, c0c is not None, pre_padded)
        return y, hn, cn

    @staticmethod
    def backward(ctx, dy, dhn, dcn):
        if dhn is not None or dcn is not None:
            raise NotImplementedError("Gradients through the final states h_n / c_n are not "
                                      "implemented (the acoustic models only use the output).")
        L = _lib.load()
        x2, w_ih_cat, w_hh, gates, csave, y, h0, c0 = ctx.saved_tensors
        if dy is None:
            dy = torch.zeros_like(y)
        pb = ctx.pb
        F, H, ndir, has_h0, has_c0, pre_padded = ctx.dims
        hprev = pb.shift(y, h0 if has_h0 else None, ndir, H)
        G4 = 4 * H
        dev = dy.device
        dy2 = dy.contiguous()
        dg = torch.empty((pb.N, ndir * G4), dtype=torch.float32, device=dev)
        state = torch.empty(L.itts_lstm_state_bytes(pb.B, H, ndir), dtype=torch.uint8, device=dev)
        want_h0, want_c0 = ctx.needs_input_grad[6], ctx.needs_input_grad[7]
        dc0_rows = torch.empty((ndir, pb.B, H), dtype=torch.float32, device=dev) if want_c0 else None
        _lib.check(L.itts_lstm_layer_bwd(_iptr(dy2), _iptr(w_hh), _iptr(c0 if has_c0 else None),
                                         _iptr(gates), _iptr(csave), pb._hptr(),
                                         _iptr(pb.d_row_off), _iptr(pb.d_rev_row), pb.T, pb.B, H, ndir,
                                         _iptr(dg), _iptr(dc0_rows), _iptr(state), ops._stream()),
                   "itts_lstm_layer_bwd")
        # trainable initial states (RNNWrapper train_hidden_init): one vector per direction shared
        # by all rows -> sum over the rows; dh0 = W_hh^T dG at the first processed frame
        dh0 = dc0 = None
        if want_h0:
            dh0 = torch.stack([ops.linear_bwd_input(
                ops.rows_gather(dg[:, d * G4:(d + 1) * G4], pb.first_rows(d)).sum(0, keepdim=True), w_hh[d])[0]
                for d in range(ndir)], dim=0)
        if want_c0:
            dc0 = dc0_rows.sum(dim=1)
        # The launch right behind a recurrence runs 20-25 % slow (the chip comes back from light load: LABNOTES 12g):
        # the two small products W_hh' take that place, the large ones follow (95.9 -> 95.6 ms per 3 x 512 BiLSTM
        # step, two A/B pairs; ITTS_RNN_BWD_SMALL_FIRST=0 for the old order)
        small_first = os.environ.get("ITTS_RNN_BWD_SMALL_FIRST", "1") != "0"
        if not small_first:
            dw_ih, db = ops.linear_bwd_weight(dg, x2)                      # [ndir*4H, F], [ndir*4H]
        dw_hh = torch.empty((ndir, G4, H), dtype=torch.float32, device=dev)
        for d in range(ndir):
            ops.linear_bwd_weight(dg[:, d * G4:(d + 1) * G4], hprev[:, d * H:(d + 1) * H],
                                  dw=dw_hh[d], want_bias=False)
        if small_first:
            dw_ih, db = ops.linear_bwd_weight(dg, x2)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops.linear_bwd_input(dg, w_ih_cat)
            if not pre_padded:
                dx = dx[:, :F]
        db = db.reshape(ndir, G4)
        return dx, None, dw_ih[:, :F].reshape(ndir, G4, F), dw_hh, db, db.clone(), dh0, dc0, None


class GRULayerFunction(torch.autograd.Function):
    """One (bi)directional GRU layer on packed rows x [N, F] (torch.nn.GRU on a PackedSequence,
    rnn_dyn/RNNWrapper.py:45-107)."""

    @staticmethod
    def forward(ctx, x2, pb, w_ih, w_hh, b_ih, b_hh, h0, training):
        L = _lib.load()
        N = x2.shape[0]
        F = w_ih.shape[-1]               # x2 may arrive with its rows already padded to 16-byte multiples
        ndir, G3, H = w_hh.shape
        T, B = pb.T, pb.B
        pre_padded = x2.shape[1] != F
        x2 = _pad4_cols(x2.contiguous())
        w_ih_cat = _pad4_cols(w_ih.reshape(ndir * G3, F))
        gin = ops.linear_fwd(x2, w_ih_cat, b_ih.reshape(-1), ops.ACT_NONE)
        dev = x2.device
        y = torch.empty((N, ndir * H), dtype=torch.float32, device=dev)
        keep = bool(training)
        gates = torch.empty((N, ndir * H * 4), dtype=torch.float32, device=dev) if keep else None
        hn = torch.empty((ndir, B, H), dtype=torch.float32, device=dev)
        state = torch.empty(L.itts_gru_state_bytes(B, H, ndir), dtype=torch.uint8, device=dev)
        w_hh_c, b_hh_c = w_hh.contiguous(), b_hh.contiguous()
        h0c = h0.contiguous() if h0 is not None else None
        _lib.check(L.itts_gru_layer_fwd(_iptr(gin), _iptr(w_hh_c), _iptr(b_hh_c), _iptr(h0c),
                                        _iptr(pb.d_lengths), pb._hptr(), _iptr(pb.d_row_off),
                                        _iptr(pb.d_rev_row), T, B, H, ndir, _iptr(y), _iptr(gates),
                                        _iptr(hn), _iptr(state), ops._stream()),
                   "itts_gru_layer_fwd")
        ctx.set_materialize_grads(False)
        if not keep:
            ctx.mark_non_differentiable(y, hn)
        if keep:
            ctx.save_for_backward(x2, w_ih_cat, w_hh_c, gates, y,
                                  h0c if h0c is not None else torch.empty(0, device=dev))
            ctx.pb = pb
            ctx.dims = (F, H, ndir, h0c is not None, pre_padded)
        return y, hn

    @staticmethod
    def backward(ctx, dy, dhn):
        if dhn is not None:
            raise NotImplementedError("Gradients through the final state h_n are not implemented "
                                      "(the acoustic models only use the output).")
        L = _lib.load()
        x2, w_ih_cat, w_hh, gates, y, h0 = ctx.saved_tensors
        if dy is None:
            dy = torch.zeros_like(y)
        pb = ctx.pb
        F, H, ndir, has_h0, pre_padded = ctx.dims
        hprev = pb.shift(y, h0 if has_h0 else None, ndir, H)
        G3 = 3 * H
        dev = dy.device
        dy2 = dy.contiguous()
        dgi = torch.empty((pb.N, ndir * G3), dtype=torch.float32, device=dev)
        dgh = torch.empty((pb.N, ndir * G3), dtype=torch.float32, device=dev)
        state = torch.empty(L.itts_gru_state_bytes(pb.B, H, ndir), dtype=torch.uint8, device=dev)
        want_h0 = ctx.needs_input_grad[6]
        dh0_rows = torch.empty((ndir, pb.B, H), dtype=torch.float32, device=dev) if want_h0 else None
        _lib.check(L.itts_gru_layer_bwd(_iptr(dy2), _iptr(w_hh), _iptr(gates),
                                        _iptr(hprev), pb._hptr(), _iptr(pb.d_row_off),
                                        _iptr(pb.d_rev_row), pb.T, pb.B, H, ndir, _iptr(dgi),
                                        _iptr(dgh), _iptr(dh0_rows), _iptr(state), ops._stream()),
                   "itts_gru_layer_bwd")
        dh0 = None
        if want_h0:      # direct part dh * z from the kernel + recurrent part W_hh^T dGh, summed over rows
            dh0 = dh0_rows.sum(dim=1) + torch.stack([ops.linear_bwd_input(
                ops.rows_gather(dgh[:, d * G3:(d + 1) * G3], pb.first_rows(d)).sum(0, keepdim=True), w_hh[d])[0]
                for d in range(ndir)], dim=0)
        small_first = os.environ.get("ITTS_RNN_BWD_SMALL_FIRST", "1") != "0"     # (as in the LSTM's backward)
        if not small_first:
            dw_ih, db_ih = ops.linear_bwd_weight(dgi, x2)                  # [ndir*3H, F], [ndir*3H]
        dw_hh = torch.empty((ndir, G3, H), dtype=torch.float32, device=dev)
        db_hh = torch.empty((ndir, G3), dtype=torch.float32, device=dev)
        for d in range(ndir):
            _, db = ops.linear_bwd_weight(dgh[:, d * G3:(d + 1) * G3], hprev[:, d * H:(d + 1) * H],
                                          dw=dw_hh[d])
            db_hh[d] = db
        if small_first:
            dw_ih, db_ih = ops.linear_bwd_weight(dgi, x2)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops.linear_bwd_input(dgi, w_ih_cat)
            if not pre_padded:
                dx = dx[:, :F]
        return dx, None, dw_ih[:, :F].reshape(ndir, G3, F), dw_hh, db_ih.reshape(ndir, G3), db_hh, \
            dh0, None
