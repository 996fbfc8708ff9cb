"""nn.Modules with torch.nn-compatible parameter names on top of the HIP kernels, so that the
reference's state dicts load unchanged (`module.0.weight`,
`module.weight_ih_l0[_reverse]`, ...)."""
import math
import os

import torch
from torch import nn

from .. import ops
from .functional import GRULayerFunction, LinearActFunction, LSTMLayerFunction, PackedBatch, StatesToCallerOrder


class LinearAct(nn.Linear):
    """torch.nn.Linear whose forward/backward run the fp32-MFMA kernels; `act` fuses the
    following Tanh / ReLU of an FFWrapper group into the GEMM epilogue."""

    def __init__(self, in_features, out_features, bias=True, act=None):
        super().__init__(in_features, out_features, bias=bias)
        self.act = ops.ACT_BY_NAME[act.lower() if isinstance(act, str) else act]

    def forward(self, input_):
        return LinearActFunction.apply(input_, self.weight, self.bias, self.act)


_cu_count = {}


def _persistent_width(H, rows, ndir, device):
    """512 when a layer of hidden size H < 512 is better run zero-padded to the width the persistent
    recurrences (csrc/rnn_persist.h) are built for, else None.  Those keep a whole recurrence inside one launch
    at ~4 us per time step where the per-step kernels other sizes take cost ~9 us, but they take 8 / ndir tiles of
    16 rows per round and the padded products are larger; measured on 3 x Bi-LSTM / GRU training steps of 2-10 s
    utterances (scripts/rnn_hidden_probe.py, profiles/r6_rnn_hidden_probe.txt): H 128 / 256 / 384 at 64
    utterances (one round) -15 / -17 / -22 % of the step, H 256 at 128 / 256 utterances 0 / +10 %, H 384 -17 / -3 %.
    ITTS_RNN_PAD_HIDDEN=0 / 1 forces it off / on; ITTS_RNN_PERSISTENT=0 (no persistent kernels) turns it off too."""
    force = os.environ.get("ITTS_RNN_PAD_HIDDEN")
    if not (128 <= H < 512) or force == "0" or os.environ.get("ITTS_RNN_PERSISTENT", "1")[:1] == "0":
        return None
    if force == "1":
        return 512
    if device.type != "cuda":
        return None
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _cu_count:
        _cu_count[idx] = torch.cuda.get_device_properties(idx).multi_processor_count
    if _cu_count[idx] != 256:
        return None
    rounds = -(-((rows + 15) // 16) // (8 // ndir))
    allowed = 1 if H < 320 else (2 if H < 448 else 4)
    return 512 if rounds <= allowed else None


def _pad_hidden(w_ih, w_hh, biases, h0s, G, H, rows=0):
    """The recurrence kernels tile the hidden units in groups of 16.  Any other hidden size runs
    zero-padded to the next multiple -- or to 512 where that puts the layer on the persistent recurrences
    (_persistent_width) --: a padded unit has zero weights and biases, so its gates sit
    at sigma(0) / tanh(0), its state stays exactly 0 and -- its W_hh columns being zero -- it never
    reaches a real unit.  Differentiable torch ops: autograd slices the gradients back.
    w_ih [ndir, G*H, F], w_hh [ndir, G*H, H], biases [ndir, G*H] each, h0s [ndir, H] or None."""
    Hp = _persistent_width(H, rows, w_ih.shape[0], w_ih.device) or (H + 15) // 16 * 16
    if Hp == H:
        return w_ih, w_hh, biases, h0s, H
    ndir, _, F = w_ih.shape
    pad = torch.nn.functional.pad
    w_ih = pad(w_ih.reshape(ndir, G, H, F), (0, 0, 0, Hp - H)).reshape(ndir, G * Hp, F)
    w_hh = pad(w_hh.reshape(ndir, G, H, H), (0, Hp - H, 0, Hp - H)).reshape(ndir, G * Hp, Hp)
    biases = [pad(b.reshape(ndir, G, H), (0, Hp - H)).reshape(ndir, G * Hp) for b in biases]
    h0s = [pad(h, (0, Hp - H)) if h is not None else None for h in h0s]
    return w_ih, w_hh, biases, h0s, Hp


def _unpad_rows(x, ndir, H, Hp):
    """[N, ndir*Hp] -> [N, ndir*H]"""
    if Hp == H:
        return x
    return x.reshape(x.shape[0], ndir, Hp)[:, :, :H].reshape(x.shape[0], ndir * H)


def _shared_initial_state(h):
    """The recurrence kernels take ONE initial state per layer and direction, shared by all rows
    (what RNNWrapper.init_hidden passes: a [.., 1, H] vector expanded over the batch).  Anything
    else would silently be replaced by row 0's state, so it is refused."""
    if h is None or h.shape[1] == 1 or h.stride(1) == 0 or getattr(h, "_itts_rows_shared", False):
        return
    if not bool((h == h[:, :1]).all()):
        raise NotImplementedError("Per-sequence initial states are not implemented: all rows of "
                                  "hx must be equal (expand one [layers*dirs, 1, H] state).")


class LSTM(nn.Module):
    """Drop-in for torch.nn.LSTM(input_size, hidden_size, num_layers, bidirectional, batch_first)
    as RNNWrapper uses it (rnn_dyn/RNNWrapper.py:45-54).  forward takes the padded tensor and
    the sequence lengths instead of a PackedSequence:
        output, (h_n, c_n) = lstm(padded, (h_0, c_0), lengths)
    with output zero-padded to the input's time extent (pad_packed_sequence(total_length=...))."""

    def __init__(self, input_size, hidden_size, num_layers=1, bias=True, batch_first=False,
                 dropout=0.0, bidirectional=False):
        super().__init__()
        assert bias, "bias=False is not supported"
        self.input_size, self.hidden_size, self.num_layers = input_size, hidden_size, num_layers
        self.batch_first, self.dropout, self.bidirectional = batch_first, dropout, bidirectional
        ndir = 2 if bidirectional else 1
        for layer in range(num_layers):
            in_size = input_size if layer == 0 else hidden_size * ndir
            for d in range(ndir):
                sfx = "_l{}{}".format(layer, "_reverse" if d == 1 else "")
                self.register_parameter("weight_ih" + sfx, nn.Parameter(torch.empty(4 * hidden_size, in_size)))
                self.register_parameter("weight_hh" + sfx, nn.Parameter(torch.empty(4 * hidden_size, hidden_size)))
                self.register_parameter("bias_ih" + sfx, nn.Parameter(torch.empty(4 * hidden_size)))
                self.register_parameter("bias_hh" + sfx, nn.Parameter(torch.empty(4 * hidden_size)))
        self.reset_parameters()

    def reset_parameters(self):
        stdv = 1.0 / math.sqrt(self.hidden_size)
        for w in self.parameters():
            nn.init.uniform_(w, -stdv, stdv)

    def _stack(self, name, layer):
        ndir = 2 if self.bidirectional else 1
        return torch.stack([getattr(self, "{}_l{}{}".format(name, layer, "_reverse" if d else ""))
                            for d in range(ndir)], dim=0)

    def forward(self, input_, hx=None, lengths=None):
        ndir = 2 if self.bidirectional else 1
        time_dim, batch_dim = (1, 0) if self.batch_first else (0, 1)
        if lengths is None:
            lengths = torch.full((input_.shape[batch_dim],), input_.shape[time_dim])
        # pack once (valid frames only, rows sorted by length), run all layers on packed rows
        pb = PackedBatch.get(lengths, input_.shape[time_dim], self.batch_first, input_.device)
        x = pb.pack(input_, pad_cols=True)
        h0 = c0 = None
        if hx is not None:
            h0, c0 = hx      # [num_layers*ndir, B, H]; RNNWrapper expands one vector per row
            _shared_initial_state(h0)
            _shared_initial_state(c0)
        hn_all, cn_all = [], []
        H = self.hidden_size
        # Every layer's operands are put together BEFORE the first recurrence is launched: the host waits for each
        # recurrence's verdict (rnn_persist.h), and what it still has to queue after that wait -- the stacking of
        # the two directions' weights, four small copies a layer -- stands between the recurrence and the next
        # layer's product on an idle device (90 us a layer boundary in the step's timeline).
        operands = []
        for layer in range(self.num_layers):
            hl = cl = None
            if h0 is not None:
                # all rows share the initial state (init_hidden expands [.., 1, H]); use row 0
                hl = h0[layer * ndir:(layer + 1) * ndir, 0, :]
                cl = c0[layer * ndir:(layer + 1) * ndir, 0, :]
            operands.append(_pad_hidden(
                self._stack("weight_ih", layer), self._stack("weight_hh", layer),
                [self._stack("bias_ih", layer), self._stack("bias_hh", layer)], [hl, cl], 4, H, rows=pb.B))
        for layer in range(self.num_layers):
            w_ih, w_hh, (b_ih, b_hh), (hl, cl), Hp = operands[layer]
            x, hn, cn = LSTMLayerFunction.apply(x, pb, w_ih, w_hh, b_ih, b_hh, hl, cl,
                                                torch.is_grad_enabled())
            x = _unpad_rows(x, ndir, H, Hp)
            if self.dropout > 0 and self.training and layer < self.num_layers - 1:
                x = torch.nn.functional.dropout(x, self.dropout, True)
            hn_all.append(hn)
            cn_all.append(cn)
        out = pb.unpack(x, input_.shape)
        # back to the caller's row order, stacked over the layers
        return out, (StatesToCallerOrder.apply(pb.inv_perm, pb.perm, self.hidden_size, *hn_all),
                     StatesToCallerOrder.apply(pb.inv_perm, pb.perm, self.hidden_size, *cn_all))


class GRU(nn.Module):
    """Drop-in for torch.nn.GRU(input_size, hidden_size, num_layers, bidirectional, batch_first)
    as RNNWrapper uses it; same parameter names (weight_ih_l0[_reverse], ...), gate order r, z, n.
        output, h_n = gru(padded, h_0, lengths)"""

    def __init__(self, input_size, hidden_size, num_layers=1, bias=True, batch_first=False,
                 dropout=0.0, bidirectional=False):
        super().__init__()
        assert bias, "bias=False is not supported"
        self.input_size, self.hidden_size, self.num_layers = input_size, hidden_size, num_layers
        self.batch_first, self.dropout, self.bidirectional = batch_first, dropout, bidirectional
        ndir = 2 if bidirectional else 1
        for layer in range(num_layers):
            in_size = input_size if layer == 0 else hidden_size * ndir
            for d in range(ndir):
                sfx = "_l{}{}".format(layer, "_reverse" if d == 1 else "")
                self.register_parameter("weight_ih" + sfx, nn.Parameter(torch.empty(3 * hidden_size, in_size)))
                self.register_parameter("weight_hh" + sfx, nn.Parameter(torch.empty(3 * hidden_size, hidden_size)))
                self.register_parameter("bias_ih" + sfx, nn.Parameter(torch.empty(3 * hidden_size)))
                self.register_parameter("bias_hh" + sfx, nn.Parameter(torch.empty(3 * hidden_size)))
        self.reset_parameters()

    def reset_parameters(self):
        stdv = 1.0 / math.sqrt(self.hidden_size)
        for w in self.parameters():
            nn.init.uniform_(w, -stdv, stdv)

    def _stack(self, name, layer):
        ndir = 2 if self.bidirectional else 1
        return torch.stack([getattr(self, "{}_l{}{}".format(name, layer, "_reverse" if d else ""))
                            for d in range(ndir)], dim=0)

    def forward(self, input_, hx=None, lengths=None):
        ndir = 2 if self.bidirectional else 1
        time_dim, batch_dim = (1, 0) if self.batch_first else (0, 1)
        if lengths is None:
            lengths = torch.full((input_.shape[batch_dim],), input_.shape[time_dim])
        pb = PackedBatch.get(lengths, input_.shape[time_dim], self.batch_first, input_.device)
        x = pb.pack(input_, pad_cols=True)
        _shared_initial_state(hx)
        hn_all = []
        H = self.hidden_size
        operands = []            # (all layers' operands before the first recurrence: see LSTM.forward)
        for layer in range(self.num_layers):
            # all rows share the initial state (init_hidden expands [.., 1, H]); use row 0
            hl = hx[layer * ndir:(layer + 1) * ndir, 0, :] if hx is not None else None
            operands.append(_pad_hidden(
                self._stack("weight_ih", layer), self._stack("weight_hh", layer),
                [self._stack("bias_ih", layer), self._stack("bias_hh", layer)], [hl], 3, H, rows=pb.B))
        for layer in range(self.num_layers):
            w_ih, w_hh, (b_ih, b_hh), (hl,), Hp = operands[layer]
            x, hn = GRULayerFunction.apply(x, pb, w_ih, w_hh, b_ih, b_hh, hl,
                                           torch.is_grad_enabled())
            x = _unpad_rows(x, ndir, H, Hp)
            if self.dropout > 0 and self.training and layer < self.num_layers - 1:
                x = torch.nn.functional.dropout(x, self.dropout, True)
            hn_all.append(hn)
        out = pb.unpack(x, input_.shape)
        return out, StatesToCallerOrder.apply(pb.inv_perm, pb.perm, self.hidden_size, *hn_all)


class RNN(nn.Module):
    """Drop-in for torch.nn.RNN(input_size, hidden_size, num_layers, nonlinearity, bidirectional,
    batch_first) as RNNWrapper builds it for 'RNNTANH' / 'RNNRELU' groups (rnn_dyn/RNNWrapper.py:
    45-54, RNNDyn.py:268-272); same parameter names.
        output, h_n = rnn(padded, h_0, lengths)
    The cell is one fused linear layer per step, h_t = act([h_{t-1} | W_ih x_t + b] [W_hh | I]^T):
    the input projection of all frames is one GEMM on packed rows, every step one launch of the
    same fp32-MFMA kernel on the rows still active (the identity block adds the projection), and
    autograd chains the steps.  Unlike LSTM / GRU there is no dedicated recurrence kernel -- this
    cell type is kept for completeness, not speed."""

    def __init__(self, input_size, hidden_size, num_layers=1, nonlinearity='tanh', bias=True,
                 batch_first=False, dropout=0.0, bidirectional=False):
        super().__init__()
        assert bias, "bias=False is not supported"
        if nonlinearity.lower() not in ("tanh", "relu"):
            raise ValueError("Unknown nonlinearity '{}'".format(nonlinearity))
        self.input_size, self.hidden_size, self.num_layers = input_size, hidden_size, num_layers
        self.nonlinearity = nonlinearity.lower()
        self.batch_first, self.dropout, self.bidirectional = batch_first, dropout, bidirectional
        ndir = 2 if bidirectional else 1
        for layer in range(num_layers):
            in_size = input_size if layer == 0 else hidden_size * ndir
            for d in range(ndir):
                sfx = "_l{}{}".format(layer, "_reverse" if d == 1 else "")
                self.register_parameter("weight_ih" + sfx, nn.Parameter(torch.empty(hidden_size, in_size)))
                self.register_parameter("weight_hh" + sfx, nn.Parameter(torch.empty(hidden_size, hidden_size)))
                self.register_parameter("bias_ih" + sfx, nn.Parameter(torch.empty(hidden_size)))
                self.register_parameter("bias_hh" + sfx, nn.Parameter(torch.empty(hidden_size)))
        self.reset_parameters()

    def reset_parameters(self):
        stdv = 1.0 / math.sqrt(self.hidden_size)
        for w in self.parameters():
            nn.init.uniform_(w, -stdv, stdv)

    def _direction(self, x, pb, layer, d, h0):
        """x [N, F] packed rows -> (y [N, H] packed, h_n [B, H] in sorted row order)"""
        H = self.hidden_size
        sfx = "_l{}{}".format(layer, "_reverse" if d == 1 else "")
        w_ih, w_hh = getattr(self, "weight_ih" + sfx), getattr(self, "weight_hh" + sfx)
        bias = getattr(self, "bias_ih" + sfx) + getattr(self, "bias_hh" + sfx)
        act = ops.ACT_TANH if self.nonlinearity == "tanh" else ops.ACT_RELU
        gin = LinearActFunction.apply(x, w_ih, bias, ops.ACT_NONE)                    # [N, H]
        w_step = torch.cat((w_hh, torch.eye(H, dtype=w_hh.dtype, device=w_hh.device)), dim=1)
        lengths = pb.h_lengths.tolist()
        row_off = pb.d_row_off.tolist()
        h = h0.unsqueeze(0).expand(pb.B, H) if h0 is not None else x.new_zeros((pb.B, H))
        steps, finished = [], []
        nact_prev = pb.B
        for s in range(pb.T):
            nact = sum(1 for n in lengths if n > s)             # sorted: the first nact rows
            if d == 0:
                g = gin[row_off[s]:row_off[s] + nact]
            else:
                g = gin.index_select(0, pb.d_rev_row[s, :nact].long())
            if nact < nact_prev:
                finished.append(h[nact:nact_prev])               # their last state is final
            h = LinearActFunction.apply(torch.cat((h[:nact], g), dim=1), w_step, None, act)
            steps.append(h)
            nact_prev = nact
        finished.append(h)
        h_n = torch.cat(finished[::-1], dim=0)                   # rows 0 .. B-1 (sorted order)
        if d == 0:
            y = torch.cat(steps, dim=0)                          # packed order is step order
        else:
            idx = torch.cat([pb.d_rev_row[s, :steps[s].shape[0]].long() for s in range(pb.T)])
            y = torch.empty_like(gin).index_copy(0, idx, torch.cat(steps, dim=0))
        return y, h_n

    def forward(self, input_, hx=None, lengths=None):
        ndir = 2 if self.bidirectional else 1
        time_dim, batch_dim = (1, 0) if self.batch_first else (0, 1)
        if lengths is None:
            lengths = torch.full((input_.shape[batch_dim],), input_.shape[time_dim])
        pb = PackedBatch.get(lengths, input_.shape[time_dim], self.batch_first, input_.device)
        x = pb.pack(input_)
        hn_all = []
        for layer in range(self.num_layers):
            outs = []
            for d in range(ndir):
                h0 = hx[layer * ndir + d, 0, :] if hx is not None else None   # shared by all rows
                y, h_n = self._direction(x, pb, layer, d, h0)
                outs.append(y)
                hn_all.append(h_n.index_select(0, pb.inv_perm))
            x = outs[0] if ndir == 1 else torch.cat(outs, dim=1)
            if self.dropout > 0 and self.training and layer < self.num_layers - 1:
                x = torch.nn.functional.dropout(x, self.dropout, True)
        return pb.unpack(x, input_.shape), torch.stack(hn_all, 0)
