"""nn.Modules with torch.nn-compatible parameter names on top of the HIP kernels, so that the
reference's state dicts load unchanged (`module.0.weight`,
`module.weight_ih_l0[_reverse]`, ...)."""
import math

import torch
from torch import nn

from .. import ops
from .functional import GRULayerFunction, LinearActFunction, LSTMLayerFunction, PackedBatch


class LinearAct(nn.Linear):
    """torch.nn.Linear whose forward/backward run the fp32-MFMA kernels; `act` fuses the
    following Tanh / ReLU of an FFWrapper group into the GEMM epilogue."""

    def __init__(self, in_features, out_features, bias=True, act=None):
        super().__init__(in_features, out_features, bias=bias)
        self.act = ops.ACT_BY_NAME[act.lower() if isinstance(act, str) else act]

    def forward(self, input_):
        return LinearActFunction.apply(input_, self.weight, self.bias, self.act)


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
        pb = PackedBatch(lengths, input_.shape[time_dim], self.batch_first, input_.device)
        x = pb.pack(input_)
        h0 = c0 = None
        if hx is not None:
            h0, c0 = hx      # [num_layers*ndir, B, H]; RNNWrapper expands one vector per row
        hn_all, cn_all = [], []
        for layer in range(self.num_layers):
            hl = cl = None
            if h0 is not None:
                # all rows share the initial state (init_hidden expands [.., 1, H]); use row 0
                hl = h0[layer * ndir:(layer + 1) * ndir, 0, :]
                cl = c0[layer * ndir:(layer + 1) * ndir, 0, :]
            x, hn, cn = LSTMLayerFunction.apply(
                x, pb, self._stack("weight_ih", layer), self._stack("weight_hh", layer),
                self._stack("bias_ih", layer), self._stack("bias_hh", layer), hl, cl,
                torch.is_grad_enabled())
            if self.dropout > 0 and self.training and layer < self.num_layers - 1:
                x = torch.nn.functional.dropout(x, self.dropout, True)
            hn_all.append(hn.index_select(1, pb.inv_perm))      # back to the caller's row order
            cn_all.append(cn.index_select(1, pb.inv_perm))
        out = pb.unpack(x, input_.shape)
        return out, (torch.cat(hn_all, 0), torch.cat(cn_all, 0))


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
        pb = PackedBatch(lengths, input_.shape[time_dim], self.batch_first, input_.device)
        x = pb.pack(input_)
        hn_all = []
        for layer in range(self.num_layers):
            # all rows share the initial state (init_hidden expands [.., 1, H]); use row 0
            hl = hx[layer * ndir:(layer + 1) * ndir, 0, :] if hx is not None else None
            x, hn = GRULayerFunction.apply(
                x, pb, self._stack("weight_ih", layer), self._stack("weight_hh", layer),
                self._stack("bias_ih", layer), self._stack("bias_hh", layer), hl,
                torch.is_grad_enabled())
            if self.dropout > 0 and self.training and layer < self.num_layers - 1:
                x = torch.nn.functional.dropout(x, self.dropout, True)
            hn_all.append(hn.index_select(1, pb.inv_perm))
        out = pb.unpack(x, input_.shape)
        return out, torch.cat(hn_all, 0)
