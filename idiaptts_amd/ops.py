"""Thin tensor-level wrappers over the C ABI (torch is used for device memory and streams only).

Every function takes CUDA(HIP) tensors, passes raw device pointers + the current HIP stream to
libidiaptts_amd.so and raises on any failure.  Nothing here computes on the CPU.
"""
import ctypes
import threading

import torch

from . import lib as _lib

ACT_NONE, ACT_TANH, ACT_RELU = 0, 1, 2
ACT_BY_NAME = {None: ACT_NONE, "none": ACT_NONE, "linear": ACT_NONE, "tanh": ACT_TANH,
               "relu": ACT_RELU}


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    # (the raw handle straight from the runtime: `torch.cuda.current_stream().cuda_stream` builds a Stream object
    # on every call, 12 us of the host's ~40 per launch on the module path)
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


def _need(t, dtype, name):
    if not t.is_cuda:
        raise _lib.IttsError("{} must live on the GPU (no CPU fallback)".format(name))
    if t.dtype != dtype:
        raise TypeError("{} must be {}, got {}".format(name, dtype, t.dtype))


def _rows(t, name):
    """2-D view contract: unit stride on the last dim, arbitrary row stride."""
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError("{} must be 2-D with contiguous rows".format(name))
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])


# ---------------------------------------------------------------------------------------- MLPG
class MlpgPlan(object):
    """What `mlpg_generation` works out from the offsets alone (checks, longest length, the one-pass kernel's table in
    launch order), prepared once for callers that solve over the same utterances again -- the streams of a batch, a
    fixed validation set: a planned call then does nothing on the host but launch (40-50 us of a 260-us lone call
    otherwise).  The scratch buffer of the largest call so far is kept as well."""

    def __init__(self, offsets):
        L = _lib.load()
        offs = _lib.offsets_array(offsets)
        self.n_utts = len(offs) - 1
        self.t_total = int(offs[self.n_utts]) if self.n_utts >= 0 else 0
        handle = ctypes.c_void_p()
        _lib.check(L.itts_mlpg_plan_create(offs, self.n_utts, ctypes.byref(handle)), "itts_mlpg_plan_create")
        self._handle = handle
        self._scratch = None

    def scratch(self, nbytes, device):
        if self._scratch is None or self._scratch.numel() < nbytes or self._scratch.device != device:
            self._scratch = torch.empty(max(nbytes, 8), dtype=torch.uint8, device=device)
        return self._scratch

    def close(self):
        handle, self._handle = self._handle, None
        if handle is not None:
            if torch.cuda.is_available():
                torch.cuda.synchronize()          # (launches read the plan's table in place)
            _lib.load().itts_mlpg_plan_destroy(handle)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def mlpg_generation(feat, variances, dim, offsets, col0=0, out=None, ocol0=0, plan=None):
    """Batched MLPG (misc/mlpg.py:94-127). feat [Ttot, >=col0+3*dim] f64 -- or f32, the acoustic model's own output
    type: widened in the solve's loads, the result is that of the f64 rows --, variances [3*dim] f64,
    offsets: python list of U+1 frame offsets (ignored with `plan`, an MlpgPlan made from them). Returns out
    [Ttot, dim] f64 (or writes into out)."""
    L = _lib.load()
    f32 = feat.dtype == torch.float32
    _need(feat, torch.float32 if f32 else torch.float64, "feat")
    _need(variances, torch.float64, "variances")
    ld = _rows(feat, "feat")
    t_total = plan.t_total if plan is not None else int(offsets[-1])
    if feat.shape[0] != t_total:
        raise ValueError("offsets[-1] != number of rows")
    if out is None:
        out = torch.empty((t_total, dim), dtype=torch.float64, device=feat.device)
    _need(out, torch.float64, "out")
    ldo = _rows(out, "out")
    nbytes = (L.itts_mlpg_scratch_bytes_f32 if f32 else L.itts_mlpg_scratch_bytes)(t_total, dim)
    if not variances.is_contiguous():
        variances = variances.contiguous()
    if plan is not None:
        scratch = plan.scratch(nbytes, feat.device)
        _lib.check(L.itts_mlpg_generation_planned(plan._handle, _ptr(feat), 1 if f32 else 0, ld, col0, dim,
                                                  _ptr(variances), _ptr(out), ldo, ocol0, _ptr(scratch), _stream()),
                   "itts_mlpg_generation_planned")
        return out
    scratch = torch.empty(max(nbytes, 8), dtype=torch.uint8, device=feat.device)
    offs = _lib.offsets_array(offsets)
    entry = L.itts_mlpg_generation_f32 if f32 else L.itts_mlpg_generation
    _lib.check(entry(_ptr(feat), ld, col0, dim, _ptr(variances),
                     offs, len(offsets) - 1, _ptr(out), ldo, ocol0,
                     _ptr(scratch), _stream()), "itts_mlpg_generation")
    return out


def lf0_vuv(f0, offsets, f0_silence_threshold=30.0, lf0_zero=0.0):
    """f0 [Ttot] f64 -> (lf0 [Ttot] f32 interpolated, vuv [Ttot] f32): WorldFeatLabelGen.py:798-802
    (float32 log, threshold, interpolate_lin) per utterance."""
    L = _lib.load()
    _need(f0, torch.float64, "f0")
    f0 = f0.contiguous()
    lf0 = torch.empty(f0.shape[0], dtype=torch.float32, device=f0.device)
    vuv = torch.empty_like(lf0)
    _lib.check(L.itts_lf0_vuv(_ptr(f0), _lib.offsets_array(offsets), len(offsets) - 1,
                              float(f0_silence_threshold), float(lf0_zero), _ptr(lf0), _ptr(vuv),
                              _stream()), "itts_lf0_vuv")
    return lf0, vuv


def sqrt_inplace(x):
    """x <- sqrt(x) for a contiguous float64 tensor (IEEE square roots: numpy's bits)."""
    L = _lib.load()
    _need(x, torch.float64, "x")
    assert x.is_contiguous()
    _lib.check(L.itts_sqrt_inplace_f64(_ptr(x), x.numel(), _stream()), "itts_sqrt_inplace_f64")
    return x


def square_inplace(x):
    """x <- x * x for a contiguous float64 tensor (one IEEE product per value: numpy's bits)."""
    L = _lib.load()
    _need(x, torch.float64, "x")
    assert x.is_contiguous()
    _lib.check(L.itts_square_inplace_f64(_ptr(x), x.numel(), _stream()), "itts_square_inplace_f64")
    return x


def interpolate_lin_f32(x, offsets):
    """interpolate_lin (misc/utils.py:40-86) on float32 contours stored back to back:
    (interpolated [Ttot], vuv [Ttot])."""
    L = _lib.load()
    _need(x, torch.float32, "x")
    x = x.contiguous()
    ip = torch.empty_like(x)
    vuv = torch.empty_like(x)
    _lib.check(L.itts_interpolate_lin_f32(_ptr(x), _lib.offsets_array(offsets), len(offsets) - 1,
                                          _ptr(ip), _ptr(vuv), _stream()),
               "itts_interpolate_lin_f32")
    return ip, vuv


def assemble_cmp(sp, lf0, vuv, bap, offsets, add_deltas=True, out=None):
    """[sp, d, dd | lf0, d, dd | vuv | bap, d, dd] (or the static layout) per frame, float32
    (save_output WorldFeatLabelGen.py:1121-1172 + compute_deltas misc/utils.py:103-105)."""
    L = _lib.load()
    for t, n in ((sp, "sp"), (lf0, "lf0"), (vuv, "vuv"), (bap, "bap")):
        _need(t, torch.float32, n)
    n_sp, n_bap = sp.shape[1], bap.shape[1]
    width = (3 if add_deltas else 1) * (n_sp + 1 + n_bap) + 1
    if out is None:
        out = torch.empty((sp.shape[0], width), dtype=torch.float32, device=sp.device)
    _lib.check(L.itts_assemble_cmp_f32(_ptr(sp), _rows(sp, "sp"), n_sp, _ptr(lf0.contiguous()),
                                       _ptr(vuv.contiguous()), _ptr(bap), _rows(bap, "bap"), n_bap,
                                       _lib.offsets_array(offsets), len(offsets) - 1,
                                       1 if add_deltas else 0, _ptr(out), _rows(out, "out"),
                                       _stream()), "itts_assemble_cmp_f32")
    return out


def feature_stats(x, col0, width, want_cov, sums=None, second=None):
    """sum x [width] and sum x x^T [width, width] (want_cov) or sum x^2 [width] over the rows of
    x[:, col0:col0+width] in fp64; adds to `sums` / `second` when given."""
    L = _lib.load()
    _need(x, torch.float32, "x")
    accumulate = sums is not None
    if sums is None:
        sums = torch.empty(width, dtype=torch.float64, device=x.device)
        second = torch.empty((width, width) if want_cov else (width,), dtype=torch.float64,
                             device=x.device)
    ws = _workspace(L.itts_feature_stats_workspace_bytes(width, 1 if want_cov else 0), x.device)
    _lib.check(L.itts_feature_stats(_ptr(x), _rows(x, "x"), x.shape[0], col0, width,
                                    1 if want_cov else 0, 1 if accumulate else 0, _ptr(sums),
                                    _ptr(second), _ptr(ws), _stream()), "itts_feature_stats")
    return sums, second


# ------------------------------------------------------------------------------ dense layers
def linear_fwd(x, w, b, act=ACT_NONE, out=None):
    L = _lib.load()
    _need(x, torch.float32, "x")
    _need(w, torch.float32, "w")
    M, K = x.shape
    N = w.shape[0]
    if w.shape[1] != K or not w.is_contiguous():
        raise ValueError("w must be contiguous [N, K]")
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=x.device)
    _lib.check(L.itts_linear_fwd(_ptr(x), _rows(x, "x"), _ptr(w), _ptr(b), _ptr(out),
                                 _rows(out, "out"), M, N, K, act, _stream()), "itts_linear_fwd")
    return out


def linear_fwd_mse(x, w, b, target, row_valid, n_valid, loss_weight=1.0, grad=None):
    """Output layer + masked MSE ('mean_per_frame') of a training step in one launch:
    (loss [1], d loss / d output [M, N]); the layer's output itself is not materialised."""
    L = _lib.load()
    _need(x, torch.float32, "x")
    _need(w, torch.float32, "w")
    _need(target, torch.float32, "target")
    _need(row_valid, torch.uint8, "row_valid")
    M, K = x.shape
    N = w.shape[0]
    if w.shape[1] != K or not w.is_contiguous():
        raise ValueError("w must be contiguous [N, K]")
    loss = torch.empty((1,), dtype=torch.float32, device=x.device)
    if grad is None:
        grad = torch.zeros((M, (N + 3) // 4 * 4), dtype=torch.float32, device=x.device)[:, :N]
    ws = _workspace(max(L.itts_linear_fwd_mse_workspace_bytes(M, N), 8), x.device)
    _lib.check(L.itts_linear_fwd_mse(_ptr(x), _rows(x, "x"), _ptr(w), _ptr(b), _ptr(target),
                                     _rows(target, "target"), _ptr(row_valid), float(n_valid),
                                     float(loss_weight), M, N, K, _ptr(loss), _ptr(grad),
                                     _rows(grad, "grad"), _ptr(ws), _stream()),
               "itts_linear_fwd_mse")
    return loss, grad


def _full_rows(t):
    """The [M, pitch] tensor a column-slice view t = buf[:, :N] was cut from (None when t is not one)."""
    if t.dim() != 2 or t.shape[0] == 0 or t.stride(1) != 1 or t.stride(0) <= t.shape[1]:
        return None
    M, P = t.shape[0], t.stride(0)
    if t.storage_offset() + M * P > t.untyped_storage().nbytes() // t.element_size():
        return None
    return torch.as_strided(t, (M, P), (P, 1))


def act_bwd(dy, y, act, out=None):
    """dz = dy * act'(y).  Rows with a padded pitch (views buf[:, :N] of [M, pitch] buffers whose pad columns
    are zero, as LinearActFunction makes them for N % 4 != 0) keep their pitch: the kernel runs over the
    whole buffers (0 * act'(0) = 0 in the pad columns) and the result is the same kind of view, so the GEMMs
    behind it still see 16-byte rows."""
    L = _lib.load()
    if out is None and dy.shape == y.shape:
        fdy, fy = _full_rows(dy), _full_rows(y)
        if fdy is not None and fy is not None and fdy.shape == fy.shape:
            full = torch.empty_like(fdy)
            _lib.check(L.itts_act_bwd(_ptr(fdy), _ptr(fy), _ptr(full), fdy.numel(), act, _stream()),
                       "itts_act_bwd")
            return full[:, :dy.shape[1]]
    dy = dy.contiguous()
    y = y.contiguous()
    if out is None:
        out = torch.empty_like(dy)
    _lib.check(L.itts_act_bwd(_ptr(dy), _ptr(y), _ptr(out), dy.numel(), act, _stream()),
               "itts_act_bwd")
    return out


def rows_gather(src, idx, width=None, fill_row=None, out=None, out_width=None):
    """out[r, :width] = src[idx[r], :width] (the fill row, or zeros, where idx[r] is outside
    [0, len(src))); columns width .. out_width - 1 are zeroed.  src: float32 [R, >= width] with unit
    column stride (any row pitch: column slices of a wider tensor are fine), idx: int64 [n_out]."""
    L = _lib.load()
    assert src.dtype == torch.float32 and src.dim() == 2 and (src.shape[1] <= 1 or src.stride(1) == 1)
    assert idx.dtype == torch.int64 and idx.is_contiguous()
    width = src.shape[1] if width is None else int(width)
    out_width = width if out_width is None else int(out_width)
    n_out = idx.numel()
    if out is None:
        out = torch.empty((n_out, out_width), dtype=torch.float32, device=src.device)
    assert out.dtype == torch.float32 and out.dim() == 2 and out.shape[0] == n_out and \
        (out.shape[1] <= 1 or out.stride(1) == 1) and out.shape[1] >= out_width
    ld_src = src.stride(0) if src.shape[0] > 1 else max(src.shape[1], 1)
    ld_dst = out.stride(0) if n_out > 1 else max(out.shape[1], 1)
    fill = fill_row.contiguous() if fill_row is not None else None
    _lib.check(L.itts_rows_gather_f32(_ptr(src), ld_src, src.shape[0], _ptr(idx), n_out, width,
                                      _ptr(fill) if fill is not None else None, _ptr(out), ld_dst, out_width,
                                      _stream()), "itts_rows_gather_f32")
    return out


def _batch_rows_2d(t, name):
    """[B, T, D] / [T, B, D] padded batch (contiguous positions) -> row pitch"""
    if t.dim() != 3 or (t.shape[2] > 1 and t.stride(2) != 1):
        raise ValueError("{} must be a 3-D padded batch with contiguous rows".format(name))
    ld = max(t.stride(1), t.shape[2], 1)
    if t.shape[0] > 1 and t.shape[1] > 0 and t.stride(0) != ld * t.shape[1]:
        raise ValueError("{}: the positions of the padded batch must be evenly spaced".format(name))
    return ld


def batch_pad_gather(rows, starts, lens, n_utts, t_max, batch_first, fill_row=None, want_mask=False, width=None,
                     rep_pos=-1, rep_row=None):
    """The padded batch of utterances whose rows lie back to back in `rows` (float32 [R, >= width]; utterance b:
    rows starts[b] .. starts[b] + lens[b] - 1; starts / lens int64 [n_utts] on the device): position (b, t) takes
    row starts[b] + t for t < lens[b], else the fill row (zeros when None) -- the padding position with flat index
    `rep_pos` takes `rep_row` instead.  Returns (padded [B, T, width] or
    [T, B, width], mask [B, T, 1] / [T, B, 1] or None) -- prepare_batch's pad_sequence + sequence_mask
    (ModularModelHandlerPyTorch.py:388-491) in one launch."""
    L = _lib.load()
    _need(rows, torch.float32, "rows")
    assert rows.dim() == 2 and (rows.shape[1] <= 1 or rows.stride(1) == 1)
    assert starts.dtype == torch.int64 and lens.dtype == torch.int64 and starts.is_cuda and lens.is_cuda
    assert starts.numel() >= n_utts and lens.numel() >= n_utts and starts.is_contiguous() and lens.is_contiguous()
    width = rows.shape[1] if width is None else int(width)
    shape = (n_utts, int(t_max), width) if batch_first else (int(t_max), n_utts, width)
    out = torch.empty(shape, dtype=torch.float32, device=rows.device)
    mask = torch.empty(shape[:2] + (1,), dtype=torch.float32, device=rows.device) if want_mask else None
    fill = fill_row.contiguous() if fill_row is not None else None
    rep = rep_row.contiguous() if rep_row is not None and rep_pos >= 0 else None
    ld_src = rows.stride(0) if rows.shape[0] > 1 else max(rows.shape[1], 1)
    _lib.check(L.itts_batch_pad_gather_f32(_ptr(rows), ld_src, rows.shape[0], _ptr(starts), _ptr(lens), n_utts,
                                           int(t_max), width, 1 if batch_first else 0,
                                           _ptr(fill) if fill is not None else None,
                                           int(rep_pos) if rep is not None else -1, _ptr(rep), _ptr(out), max(width, 1),
                                           _ptr(mask) if mask is not None else None, _stream()),
               "itts_batch_pad_gather_f32")
    return out, mask


def batch_pack_rows(padded, starts, lens, batch_first, n_rows, out_width=None, out=None, rep_pos=-1, rep_dst_row=-1):
    """rows[starts[b] + t, :] = padded position (b, t) for t < lens[b] (the adjoint of batch_pad_gather); columns
    width .. out_width - 1 of the written rows are zeroed.  `n_rows` rows are allocated (rows no valid position maps
    to are left unwritten: the caller owns them).  rep_pos >= 0: that padding position is copied to row rep_dst_row."""
    L = _lib.load()
    _need(padded, torch.float32, "padded")
    ld = _batch_rows_2d(padded, "padded")
    n_utts, t_max = (padded.shape[0], padded.shape[1]) if batch_first else (padded.shape[1], padded.shape[0])
    width = padded.shape[2]
    out_width = width if out_width is None else int(out_width)
    if out is None:
        out = torch.empty((int(n_rows), out_width), dtype=torch.float32, device=padded.device)
    assert out.dim() == 2 and out.shape[0] >= n_rows and out.shape[1] >= out_width and \
        (out.shape[1] <= 1 or out.stride(1) == 1)
    ld_dst = out.stride(0) if out.shape[0] > 1 else max(out.shape[1], 1)
    _lib.check(L.itts_batch_pack_rows_f32(_ptr(padded), ld, _ptr(starts), _ptr(lens), n_utts, t_max, width,
                                          1 if batch_first else 0, _ptr(out), ld_dst, out_width, int(rep_pos),
                                          int(rep_dst_row), _stream()),
               "itts_batch_pack_rows_f32")
    return out


def batch_concat_rows(rows, table, n_out, t_max, width=None, out_width=None):
    """out[dst_starts[b] + t] = rows[src_starts[b] + t] for t < lens[b]; table: int64 [3, B] on the device (src starts,
    dst starts, lens); n_out rows of out_width floats (columns beyond `width` zero).  FrameShard.gather's batches."""
    L = _lib.load()
    _need(rows, torch.float32, "rows")
    assert rows.dim() == 2 and (rows.shape[1] <= 1 or rows.stride(1) == 1)
    assert table.dtype == torch.int64 and table.is_cuda and table.dim() == 2 and table.shape[0] == 3 and \
        table.is_contiguous()
    width = rows.shape[1] if width is None else int(width)
    out_width = width if out_width is None else int(out_width)
    out = torch.empty((int(n_out), out_width), dtype=torch.float32, device=rows.device)
    ld_src = rows.stride(0) if rows.shape[0] > 1 else max(rows.shape[1], 1)
    _lib.check(L.itts_batch_concat_rows_f32(_ptr(rows), ld_src, rows.shape[0], _ptr(table[0]), _ptr(table[1]),
                                            _ptr(table[2]), table.shape[1], int(t_max), width, _ptr(out),
                                            max(out_width, 1), out_width, _stream()), "itts_batch_concat_rows_f32")
    return out


def batch_pad_colsum(padded, lens, batch_first, out=None):
    """Per column, the sum of a padded batch over its padding positions (t >= lens[b]), in a fixed order; into `out`
    (contiguous, at least as wide: the further columns are zeroed) when given."""
    L = _lib.load()
    _need(padded, torch.float32, "padded")
    ld = _batch_rows_2d(padded, "padded")
    n_utts, t_max = (padded.shape[0], padded.shape[1]) if batch_first else (padded.shape[1], padded.shape[0])
    width = padded.shape[2]
    if out is None:
        out = torch.empty(width, dtype=torch.float32, device=padded.device)
    assert out.dim() == 1 and out.is_contiguous() and out.numel() >= width
    ws = torch.empty(max(int(L.itts_batch_pad_colsum_workspace_bytes(n_utts * t_max, width)), 4), dtype=torch.uint8,
                     device=padded.device)
    _lib.check(L.itts_batch_pad_colsum_f32(_ptr(padded), ld, _ptr(lens), n_utts, t_max, width,
                                           1 if batch_first else 0, _ptr(out), out.numel(), _ptr(ws), _stream()),
               "itts_batch_pad_colsum_f32")
    return out


def linear_bwd_input(dz, w, yprev=None, act_prev=ACT_NONE, out=None):
    L = _lib.load()
    _need(dz, torch.float32, "dz")
    M, N = dz.shape
    K = w.shape[1]
    if out is None:
        out = torch.empty((M, K), dtype=torch.float32, device=dz.device)
    _lib.check(L.itts_linear_bwd_input(_ptr(dz), _rows(dz, "dz"), _ptr(w), _ptr(out),
                                       _rows(out, "out"), _ptr(yprev),
                                       _rows(yprev, "yprev") if yprev is not None else 0,
                                       act_prev, M, N, K, _stream()), "itts_linear_bwd_input")
    return out


_ws_cache = {}
class _DeferState(threading.local):
    """Per host thread, like the library's own deferral switch (thread_local on the C side): calls made
    from another thread -- the autograd worker -- are neither deferred nor counted here."""
    on = False
    next = 0


_defer = _DeferState()


def _workspace(nbytes, device):
    """Scratch of the current (device, stream): two streams never share a workspace.  Inside
    `deferred_reductions()` every call gets a workspace of its own (the partial results it leaves
    there are read by the one reduction launch at the end)."""
    key = (device.index if device.index is not None else torch.cuda.current_device(),
           torch.cuda.current_stream(device).cuda_stream)
    if _defer.on:
        key = key + (_defer.next,)
        _defer.next += 1
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


class deferred_reductions:
    """with ops.deferred_reductions(): the loss and gradient entry points called inside leave their
    partial sums / split-K slabs in place, and ONE launch reduces them all when the block ends
    (itts_defer_reductions / itts_reduce_deferred; results bit-identical to the separate launches).
    Nothing inside the block may read a loss or a weight / bias gradient produced inside it.
    Blocks do not nest (the inner block's workspaces would be the outer block's still-pending slabs)."""

    def __enter__(self):
        if _defer.on:
            raise RuntimeError("deferred_reductions() blocks do not nest")
        _lib.check(_lib.load().itts_defer_reductions(1), "itts_defer_reductions")
        _defer.on, _defer.next = True, 0
        return self

    def __exit__(self, exc_type, exc, tb):
        _defer.on = False
        _lib.check(_lib.load().itts_reduce_deferred(_stream()), "itts_reduce_deferred")
        return False


def linear_bwd_weight(dz, x, dw=None, db=None, accumulate=False, want_bias=True):
    L = _lib.load()
    _need(dz, torch.float32, "dz")
    _need(x, torch.float32, "x")
    M, N = dz.shape
    K = x.shape[1]
    if dw is None:
        dw = torch.empty((N, K), dtype=torch.float32, device=dz.device)
    if db is None and want_bias:
        db = torch.empty((N,), dtype=torch.float32, device=dz.device)
    ws = _workspace(L.itts_linear_bwd_weight_workspace_bytes(M, N, K), dz.device)
    _lib.check(L.itts_linear_bwd_weight(_ptr(dz), _rows(dz, "dz"), _ptr(x), _rows(x, "x"),
                                        _ptr(dw), _ptr(db), M, N, K, _ptr(ws),
                                        1 if accumulate else 0, _stream()),
               "itts_linear_bwd_weight")
    return dw, db


def linear_bwd(dz, x, w, dw, db, dx, yprev=None, act_prev=ACT_NONE, accumulate=False):
    """dw [N, K] (+ db [N]) and dx [M, K] of a linear layer from dz [M, N], its input x [M, K] and its
    weight w [N, K] in one call (one launch when the buffers allow 16-byte rows); dx is multiplied by
    the derivative of the previous layer's activation when yprev (that layer's output) is given."""
    L = _lib.load()
    _need(dz, torch.float32, "dz")
    _need(x, torch.float32, "x")
    _need(w, torch.float32, "w")
    M, N = dz.shape
    K = x.shape[1]
    if w.shape != (N, K) or not w.is_contiguous() or not dw.is_contiguous() or dw.shape != (N, K):
        raise ValueError("w and dw must be contiguous [N, K]")
    ws = _workspace(L.itts_linear_bwd_weight_workspace_bytes(M, N, K), dz.device)
    _lib.check(L.itts_linear_bwd(_ptr(dz), _rows(dz, "dz"), _ptr(x), _rows(x, "x"), _ptr(w), _ptr(dw),
                                 _ptr(db), _ptr(dx), _rows(dx, "dx"), _ptr(yprev),
                                 _rows(yprev, "yprev") if yprev is not None else 0,
                                 int(act_prev if act_prev is not None else ACT_NONE), M, N, K, _ptr(ws),
                                 1 if accumulate else 0, _stream()), "itts_linear_bwd")
    return dw, db, dx


def masked_mse(pred, target, row_valid, n_valid, loss_weight=1.0, want_grad=True, grad=None):
    """NamedLoss(MSELoss, 'mean_per_frame') (loss/NamedLoss.py:70-117) on [M, D] rows."""
    L = _lib.load()
    _need(pred, torch.float32, "pred")
    _need(target, torch.float32, "target")
    _need(row_valid, torch.uint8, "row_valid")
    M, D = pred.shape
    loss = torch.empty((1,), dtype=torch.float32, device=pred.device)
    if want_grad and grad is None:
        grad = torch.empty((M, D), dtype=torch.float32, device=pred.device)
    ws = _workspace(max(L.itts_masked_mse_workspace_bytes(M, D), 8), pred.device)
    _lib.check(L.itts_masked_mse(_ptr(pred), _rows(pred, "pred"), _ptr(target),
                                 _rows(target, "target"), _ptr(row_valid), M, D, float(n_valid),
                                 float(loss_weight), _ptr(loss),
                                 _ptr(grad) if want_grad else None,
                                 _rows(grad, "grad") if want_grad else 0, _ptr(ws), _stream()),
               "itts_masked_mse")
    return loss, (grad if want_grad else None)


def weighted_loss(pred, target, row_weight, kind, want_grad=True, want_elem=False):
    """sum_r w[r] sum_c e(pred - target) on [M, D] rows, e squared (kind 0) / absolute (kind 1)
    error: (loss [1], grad or None, elementwise values or None)."""
    L = _lib.load()
    _need(pred, torch.float32, "pred")
    _need(target, torch.float32, "target")
    _need(row_weight, torch.float32, "row_weight")
    M, D = pred.shape
    loss = torch.empty((1,), dtype=torch.float32, device=pred.device)
    grad = torch.empty((M, D), dtype=torch.float32, device=pred.device) if want_grad else None
    elem = torch.empty((M, D), dtype=torch.float32, device=pred.device) if want_elem else None
    ws = torch.empty(max(L.itts_masked_mse_workspace_bytes(M, D), 8), dtype=torch.uint8,
                     device=pred.device)
    _lib.check(L.itts_weighted_loss(_ptr(pred), _rows(pred, "pred"), _ptr(target),
                                    _rows(target, "target"), _ptr(row_weight.contiguous()), M, D,
                                    int(kind), _ptr(loss), _ptr(grad), D if want_grad else 0,
                                    _ptr(elem), D if want_elem else 0, _ptr(ws), _stream()),
               "itts_weighted_loss")
    return loss, grad, elem


def adam_step(param, grad, exp_avg, exp_avg_sq, step, lr=1e-3, betas=(0.9, 0.999), eps=1e-8,
              weight_decay=0.0, grad_scale=1.0):
    L = _lib.load()
    for t, n in ((param, "param"), (grad, "grad"), (exp_avg, "exp_avg"),
                 (exp_avg_sq, "exp_avg_sq")):
        _need(t, torch.float32, n)
        if not t.is_contiguous():
            raise ValueError(n + " must be contiguous")
    _lib.check(L.itts_adam_step(_ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq),
                                param.numel(), lr, betas[0], betas[1], eps, weight_decay,
                                int(step), grad_scale, _stream()), "itts_adam_step")


def grad_norm_accum(x, accum, norm_kind=2, accumulate=False):
    """accum[0] (+)= sum x^2 (norm_kind 2) or max(accum[0], max |x|) (norm_kind 0 = infinity norm)
    of a flat f32 buffer; the input of the clipping inside adam_step_fused."""
    L = _lib.load()
    _need(x, torch.float32, "x")
    _need(accum, torch.float32, "accum")
    if not x.is_contiguous():
        raise ValueError("x must be contiguous")
    _lib.check(L.itts_grad_norm_accum(_ptr(x), x.numel(), int(norm_kind), _ptr(accum),
                                      int(bool(accumulate)), _ptr(_workspace(4096, x.device)),
                                      _stream()), "itts_grad_norm_accum")
    return accum


def adam_step_fused(param, grad, exp_avg, exp_avg_sq, step, lr=1e-3, betas=(0.9, 0.999), eps=1e-8,
                    weight_decay=0.0, grad_scale=1.0, norm_accum=None, norm_kind=2,
                    clip_max_norm=0.0, clip_value=0.0, ema_shadow=None, ema_decay=0.0):
    """Gradient clipping (by norm from `norm_accum`, by value), Adam and the parameter EMA in one
    pass over flat f32 buffers."""
    L = _lib.load()
    for t, n in ((param, "param"), (grad, "grad"), (exp_avg, "exp_avg"),
                 (exp_avg_sq, "exp_avg_sq")) + \
            (((ema_shadow, "ema_shadow"),) if ema_shadow is not None else ()):
        _need(t, torch.float32, n)
        if not t.is_contiguous() or t.numel() != param.numel():
            raise ValueError(n + " must be contiguous and as long as param")
    _lib.check(L.itts_adam_step_fused(
        _ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), param.numel(), lr, betas[0],
        betas[1], eps, weight_decay, int(step), grad_scale,
        _ptr(norm_accum) if norm_accum is not None else None, int(norm_kind),
        float(clip_max_norm), float(clip_value or 0.0),
        _ptr(ema_shadow) if ema_shadow is not None else None, float(ema_decay), _stream()),
        "itts_adam_step_fused")


def sgd_step(param, grad, momentum_buf=None, first_step=False, lr=1e-3, momentum=0.0,
             dampening=0.0, weight_decay=0.0, nesterov=False, grad_scale=1.0):
    L = _lib.load()
    for t, n in ((param, "param"), (grad, "grad")) + \
            (((momentum_buf, "momentum_buf"),) if momentum_buf is not None else ()):
        _need(t, torch.float32, n)
        if not t.is_contiguous():
            raise ValueError(n + " must be contiguous")
    _lib.check(L.itts_sgd_step(_ptr(param), _ptr(grad),
                               _ptr(momentum_buf) if momentum_buf is not None else None,
                               param.numel(), lr, momentum, dampening, weight_decay,
                               int(bool(nesterov)), int(bool(first_step)), grad_scale,
                               _stream()), "itts_sgd_step")


def ema_update(shadow, param, decay):
    L = _lib.load()
    for t, n in ((shadow, "shadow"), (param, "param")):
        _need(t, torch.float32, n)
        if not t.is_contiguous():
            raise ValueError(n + " must be contiguous")
    _lib.check(L.itts_ema_update(_ptr(shadow), _ptr(param), shadow.numel(), float(decay),
                                 _stream()), "itts_ema_update")


# ------------------------------------------------------------------------- WORLD frame kernels
def _c_int_ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def cheaptrick_mcep(x, x_off, f0, f_off, fs, frame_period=5.0, fft_size=None, q1=-0.15,
                    want_sp=True, order=None, alpha=None, eps=1e-8, miniter=2, maxiter=30,
                    threshold=1e-3, mc_dtype=torch.float32, want_iters=False):
    """CheapTrick (+ fused SPTK mcep when `order` is given) for utterances stored back to back.
    x f64 [Ntot], f0 f64 [Ttot]; x_off / f_off python lists (U+1). Returns (sp or None,
    mcep or None, iters or None)."""
    L = _lib.load()
    _need(x, torch.float64, "x")
    _need(f0, torch.float64, "f0")
    fft_size = fft_size or L.itts_cheaptrick_fft_size(fs, 71.0)
    T = int(f_off[-1])
    K = fft_size // 2 + 1
    sp = torch.empty((T, K), dtype=torch.float64, device=x.device) if want_sp else None
    mc32 = mc64 = iters = None
    if order is not None:
        if mc_dtype == torch.float32:
            mc32 = torch.empty((T, order + 1), dtype=torch.float32, device=x.device)
        else:
            mc64 = torch.empty((T, order + 1), dtype=torch.float64, device=x.device)
        if want_iters:
            iters = torch.empty((T,), dtype=torch.int32, device=x.device)
    _lib.check(L.itts_cheaptrick_mcep(_ptr(x), _lib.offsets_array(x_off), _ptr(f0),
                                      _lib.offsets_array(f_off), len(x_off) - 1, fs,
                                      float(frame_period), fft_size, q1, _ptr(sp),
                                      order if order is not None else 0,
                                      float(alpha) if alpha is not None else 0.0, eps, miniter,
                                      maxiter, threshold, _ptr(mc32), (order or 0) + 1, _ptr(mc64),
                                      _ptr(iters), _stream()), "itts_cheaptrick_mcep")
    return sp, (mc32 if mc32 is not None else mc64), iters


def mcep(amp_sp, order, alpha, eps=1e-8, miniter=2, maxiter=30, threshold=1e-3,
         dtype=torch.float32, want_iters=False):
    """pysptk.mcep(amp_sp, order, alpha, eps, etype=1, itype=3) on [T, K] f64 (GPU)."""
    L = _lib.load()
    _need(amp_sp, torch.float64, "amp_sp")
    amp_sp = amp_sp.contiguous()
    T, K = amp_sp.shape
    out = torch.empty((T, order + 1), dtype=dtype, device=amp_sp.device)
    iters = torch.empty((T,), dtype=torch.int32, device=amp_sp.device) if want_iters else None
    f32 = out if dtype == torch.float32 else None
    f64 = out if dtype == torch.float64 else None
    _lib.check(L.itts_mcep(_ptr(amp_sp), T, K, order, float(alpha), eps, miniter, maxiter,
                           threshold, _ptr(f32), order + 1, _ptr(f64), _ptr(iters), _stream()),
               "itts_mcep")
    return (out, iters) if want_iters else out


def mgc2sp(mc, alpha, fftlen, want_logamp=False, want_pow=False):
    """exp(float32(pysptk.mgc2sp(mc, alpha, 0, fftlen).real)) -> [T, fftlen/2+1] f32
    (or the f64 log amplitude when want_logamp, or the f64 power spectrum when want_pow)."""
    L = _lib.load()
    _need(mc, torch.float64, "mc")
    mc = mc.contiguous()
    T, m1 = mc.shape
    K = fftlen // 2 + 1
    out32 = None if (want_logamp or want_pow) else torch.empty((T, K), dtype=torch.float32,
                                                               device=mc.device)
    out64 = torch.empty((T, K), dtype=torch.float64, device=mc.device) if want_logamp else None
    outpw = torch.empty((T, K), dtype=torch.float64, device=mc.device) if want_pow else None
    _lib.check(L.itts_mgc2sp(_ptr(mc), T, m1 - 1, float(alpha), fftlen, _ptr(out32), _ptr(out64),
                             _ptr(outpw), _stream()), "itts_mgc2sp")
    return out64 if want_logamp else (outpw if want_pow else out32)


def mgcep(sp, order, alpha, gamma, eps=1e-8, miniter=2, maxiter=30, threshold=1e-3,
          dtype=torch.float32, want_iters=False, input_is_power=False):
    """pysptk.mgcep(amp_sp, order, alpha, gamma, eps, min_det=0, etype=1, itype=3) on [T, K] f64
    amplitude spectra (or power spectra with input_is_power)."""
    L = _lib.load()
    _need(sp, torch.float64, "sp")
    sp = sp.contiguous()
    T, K = sp.shape
    out = torch.empty((T, order + 1), dtype=dtype, device=sp.device)
    iters = torch.empty((T,), dtype=torch.int32, device=sp.device) if want_iters else None
    f32 = out if dtype == torch.float32 else None
    f64 = out if dtype == torch.float64 else None
    _lib.check(L.itts_mgcep(_ptr(sp), 1 if input_is_power else 0, T, K, order, float(alpha),
                            float(gamma), eps, miniter, maxiter, threshold, _ptr(f32), order + 1,
                            _ptr(f64), _ptr(iters), _stream()), "itts_mgcep")
    return (out, iters) if want_iters else out


def mgc2sp_gamma(mgc, alpha, gamma, fftlen, want_logamp=False, want_pow=False):
    """exp(float32(pysptk.mgc2sp(mgc, alpha, gamma, fftlen).real)) -> [T, fftlen/2+1] f32 (or the f64
    log amplitude / f64 power spectrum)."""
    L = _lib.load()
    _need(mgc, torch.float64, "mgc")
    mgc = mgc.contiguous()
    T, m1 = mgc.shape
    K = fftlen // 2 + 1
    out32 = None if (want_logamp or want_pow) else torch.empty((T, K), dtype=torch.float32,
                                                               device=mgc.device)
    out64 = torch.empty((T, K), dtype=torch.float64, device=mgc.device) if want_logamp else None
    outpw = torch.empty((T, K), dtype=torch.float64, device=mgc.device) if want_pow else None
    _lib.check(L.itts_mgc2sp_gamma(_ptr(mgc), T, m1 - 1, float(alpha), float(gamma), fftlen,
                                   _ptr(out32), _ptr(out64), _ptr(outpw), _stream()),
               "itts_mgc2sp_gamma")
    return out64 if want_logamp else (outpw if want_pow else out32)


def code_aperiodicity(ap, fs, dtype=torch.float64):
    L = _lib.load()
    _need(ap, torch.float64, "ap")
    ap = ap.contiguous()
    T, K = ap.shape
    nap = L.itts_num_aperiodicities(fs)
    out = torch.empty((T, nap), dtype=dtype, device=ap.device)
    _lib.check(L.itts_code_aperiodicity(_ptr(ap), T, (K - 1) * 2, fs,
                                        _ptr(out) if dtype == torch.float64 else None,
                                        _ptr(out) if dtype == torch.float32 else None, _stream()),
               "itts_code_aperiodicity")
    return out


def decode_aperiodicity(bap, fs, fft_size, voiced_f0=None):
    """pyworld.decode_aperiodicity: bap [T, nap] f64 -> ap [T, fft_size / 2 + 1] f64.  With `voiced_f0` (f0 [T] f64)
    only the rows the synthesis of that contour can read -- voiced frames and their neighbours -- are decoded, the
    others are uninitialised memory: for world_synthesize and nothing else."""
    L = _lib.load()
    _need(bap, torch.float64, "bap")
    bap = bap.contiguous()
    T = bap.shape[0]
    out = torch.empty((T, fft_size // 2 + 1), dtype=torch.float64, device=bap.device)
    if voiced_f0 is not None:
        _need(voiced_f0, torch.float64, "voiced_f0")
        if voiced_f0.numel() != T:
            raise ValueError("voiced_f0 and bap differ in frames")
        _lib.check(L.itts_decode_aperiodicity_voiced(_ptr(bap), _ptr(voiced_f0.contiguous()), T, fs, fft_size,
                                                     _ptr(out), _stream()), "itts_decode_aperiodicity_voiced")
        return out
    _lib.check(L.itts_decode_aperiodicity(_ptr(bap), T, fs, fft_size, _ptr(out), _stream()),
               "itts_decode_aperiodicity")
    return out


def dio(x, x_off, f_off, fs, frame_period=5.0, f0_floor=71.0, f0_ceil=800.0,
        channels_in_octave=2.0, allowed_range=0.1):
    """pyworld.dio for utterances stored back to back -> f0 [Ttot] f64."""
    L = _lib.load()
    _need(x, torch.float64, "x")
    f0 = torch.empty((int(f_off[-1]),), dtype=torch.float64, device=x.device)
    _lib.check(L.itts_dio(_ptr(x), _lib.offsets_array(x_off), _lib.offsets_array(f_off),
                          len(x_off) - 1, fs, float(frame_period), f0_floor, f0_ceil,
                          channels_in_octave, allowed_range, _ptr(f0), _stream()), "itts_dio")
    return f0


def harvest_num_frames(n_samples, fs, frame_period=5.0):
    return int(_lib.load().itts_harvest_num_frames(int(n_samples), int(fs), float(frame_period)))


def harvest(x, x_off, f_off, fs, frame_period=5.0, f0_floor=71.0, f0_ceil=800.0, stages=False):
    """pyworld.harvest for utterances stored back to back: f0 [Ttot] f64 (frame counts from
    harvest_num_frames).  stages=True (one utterance): also a dict with the raw band candidates,
    the refined candidates / scores and the 1 ms contour before smoothing."""
    L = _lib.load()
    _need(x, torch.float64, "x")
    f0 = torch.empty((int(f_off[-1]),), dtype=torch.float64, device=x.device)
    dbg = {}
    ptrs = [None, None, None, None]
    if stages:
        import math
        assert len(x_off) == 2, "stages are returned for a single utterance"
        nch = 1 + int(math.log(f0_ceil * 1.1 / (f0_floor * 0.9)) / math.log(2.0) * 40.0)
        maxc = int(nch / 10.0 + 0.5) * 7
        T1 = harvest_num_frames(int(x_off[1]) - int(x_off[0]), fs, 1.0)
        dbg = dict(raw=torch.zeros((nch, T1), dtype=torch.float64, device=x.device),
                   cand=torch.zeros((T1, maxc), dtype=torch.float64, device=x.device),
                   score=torch.zeros((T1, maxc), dtype=torch.float64, device=x.device),
                   best=torch.zeros((T1,), dtype=torch.float64, device=x.device))
        ptrs = [_ptr(dbg[k]) for k in ("raw", "cand", "score", "best")]
    _lib.check(L.itts_harvest(_ptr(x), _lib.offsets_array(x_off), _lib.offsets_array(f_off),
                              len(x_off) - 1, fs, float(frame_period), float(f0_floor),
                              float(f0_ceil), _ptr(f0), ptrs[0], ptrs[1], ptrs[2], ptrs[3],
                              _stream()), "itts_harvest")
    return (f0, dbg) if stages else f0


def wav2world(x, x_off, f_off, fs, frame_period=5.0, fft_size=None, want_sp=True, want_ap=True):
    """pyworld.wav2world for utterances stored back to back: (f0 [Ttot], sp [Ttot, K] or None,
    ap [Ttot, K] or None), all f64 (WorldFeatLabelGen.py:792-793)."""
    L = _lib.load()
    _need(x, torch.float64, "x")
    fft_size = fft_size or L.itts_cheaptrick_fft_size(fs, 71.0)
    T, K = int(f_off[-1]), fft_size // 2 + 1
    f0 = torch.empty((T,), dtype=torch.float64, device=x.device)
    sp = torch.empty((T, K), dtype=torch.float64, device=x.device) if want_sp else None
    ap = torch.empty((T, K), dtype=torch.float64, device=x.device) if want_ap else None
    _lib.check(L.itts_wav2world(_ptr(x), _lib.offsets_array(x_off), _lib.offsets_array(f_off),
                                len(x_off) - 1, fs, float(frame_period), fft_size, _ptr(f0),
                                _ptr(sp), _ptr(ap), _stream()), "itts_wav2world")
    return f0, sp, ap


def stonemask(x, x_off, f0, f_off, fs, frame_period=5.0):
    L = _lib.load()
    _need(x, torch.float64, "x")
    _need(f0, torch.float64, "f0")
    out = torch.empty_like(f0)
    _lib.check(L.itts_stonemask(_ptr(x), _lib.offsets_array(x_off), _ptr(f0),
                                _lib.offsets_array(f_off), len(x_off) - 1, fs,
                                float(frame_period), _ptr(out), _stream()), "itts_stonemask")
    return out


def d4c(x, x_off, f0, f_off, fs, frame_period=5.0, fft_size=None, threshold=0.85, want_ap=True,
        want_bap=None):
    """pyworld.d4c (+ optional fused code_aperiodicity). want_bap: None | torch.float64 | float32.
    Returns (ap or None, bap or None)."""
    L = _lib.load()
    _need(x, torch.float64, "x")
    _need(f0, torch.float64, "f0")
    fft_size = fft_size or L.itts_cheaptrick_fft_size(fs, 71.0)
    T = int(f_off[-1])
    nap = L.itts_num_aperiodicities(fs)
    ap = torch.empty((T, fft_size // 2 + 1), dtype=torch.float64, device=x.device) if want_ap \
        else None
    bap = torch.empty((T, nap), dtype=want_bap, device=x.device) if want_bap is not None else None
    _lib.check(L.itts_d4c(_ptr(x), _lib.offsets_array(x_off), _ptr(f0), _lib.offsets_array(f_off),
                          len(x_off) - 1, fs, float(frame_period), fft_size, threshold, _ptr(ap),
                          _ptr(bap) if want_bap == torch.float64 else None,
                          _ptr(bap) if want_bap == torch.float32 else None, nap, _stream()),
               "itts_d4c")
    return ap, bap


def synth_offsets(f_off, fs, frame_period=5.0):
    """Sample offsets of the synthesised utterances for the frame offsets f_off (pyworld.synthesize's output lengths)."""
    L = _lib.load()
    y_off = [0]
    for u in range(len(f_off) - 1):
        y_off.append(y_off[-1] + L.itts_world_synth_length(int(f_off[u + 1] - f_off[u]), fs,
                                                           float(frame_period)))
    return y_off


def world_synthesize(f0, sp, ap, f_off, fs, frame_period=5.0, preemphasis=0.0,
                     dtype=torch.float32, spectra_ready=None, y_off=None, ap_ready=None):
    """pyworld.synthesize + float32 cast + de-pre-emphasis for utterances stored back to back.
    Returns (y [Ytot], y_off list).  spectra_ready: a torch.cuda.Event recorded behind the producer of sp (and of ap,
    unless ap_ready is given: an event of its own behind the producer of ap) when they run on another stream: the part
    of the synthesis in front of the pulse kernels needs f0 only, the unvoiced pulses the envelope only."""
    L = _lib.load()
    for t, n in ((f0, "f0"), (sp, "sp"), (ap, "ap")):
        _need(t, torch.float64, n)
    sp = sp.contiguous()
    ap = ap.contiguous()
    K = sp.shape[1]
    fft_size = (K - 1) * 2
    if y_off is None:
        y_off = synth_offsets(f_off, fs, frame_period)
    y = torch.empty((y_off[-1],), dtype=dtype, device=f0.device)
    args = (_ptr(f0), _ptr(sp), _ptr(ap), _lib.offsets_array(f_off),
            _lib.offsets_array(y_off), len(f_off) - 1, fs,
            float(frame_period), fft_size, float(preemphasis),
            _ptr(y) if dtype == torch.float32 else None,
            _ptr(y) if dtype == torch.float64 else None, _stream())
    if spectra_ready is None:
        _lib.check(L.itts_world_synthesize(*args), "itts_world_synthesize")
    else:
        ap_ev = ap_ready if ap_ready is not None else spectra_ready
        _lib.check(L.itts_world_synthesize_after(*args, ctypes.c_void_p(spectra_ready.cuda_event),
                                                 ctypes.c_void_p(ap_ev.cuda_event)),
                   "itts_world_synthesize_after")
        cur = torch.cuda.current_stream(f0.device)
        sp.record_stream(cur)
        ap.record_stream(cur)
    return y, y_off
