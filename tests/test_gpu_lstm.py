"""GPU parity of the (Bi)LSTM path against torch.nn.LSTM on the CPU fed with
pack_padded_sequence(enforce_sorted=False) -- the exact calls of rnn_dyn/RNNWrapper.py:89-102."""
import numpy as np
import pytest
import torch
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence

pytestmark = pytest.mark.gpu


def _copy_params(src, dst):
    with torch.no_grad():
        for (n1, p1), (n2, p2) in zip(src.named_parameters(), dst.named_parameters()):
            assert n1 == n2 and p1.shape == p2.shape
            p2.copy_(p1)


@pytest.mark.parametrize("in_dim,H,layers,bidir,lengths,batch_first", [
    (20, 16, 1, False, [7], False),
    (20, 16, 1, True, [9, 4, 6], False),
    (37, 32, 2, True, [12, 12, 1, 5, 9], True),
    (425, 64, 3, True, [40, 33, 17], False),
])
def test_lstm_forward_backward_match_torch(gpu, in_dim, H, layers, bidir, lengths, batch_first):
    from idiaptts_amd.nn import LSTM
    torch.manual_seed(0)
    ref = torch.nn.LSTM(in_dim, H, layers, bidirectional=bidir, batch_first=batch_first).double()
    mine = LSTM(in_dim, H, layers, bidirectional=bidir, batch_first=batch_first)
    with torch.no_grad():
        for (n1, p1), (n2, p2) in zip(ref.named_parameters(), mine.named_parameters()):
            assert n1 == n2
            p2.copy_(p1.float())
    mine = mine.to(gpu)
    assert list(mine.state_dict().keys()) == list(ref.state_dict().keys())
    B, T = len(lengths), max(lengths)
    x = torch.randn(B, T, in_dim) if batch_first else torch.randn(T, B, in_dim)
    lt = torch.tensor(lengths)
    for b, l in enumerate(lengths):      # garbage in the padding must not matter
        if batch_first:
            x[b, l:] = 7.0
        else:
            x[l:, b] = 7.0
    ndir = 2 if bidir else 1
    h0 = torch.randn(layers * ndir, 1, H).expand(-1, B, -1).contiguous() * 0.3
    c0 = torch.randn(layers * ndir, 1, H).expand(-1, B, -1).contiguous() * 0.3
    xr = x.double().requires_grad_(True)
    packed = pack_padded_sequence(xr, lt, batch_first=batch_first, enforce_sorted=False)
    out_p, (hn_r, cn_r) = ref(packed, (h0.double(), c0.double()))
    out_r, _ = pad_packed_sequence(out_p, batch_first=batch_first, total_length=T)
    w = torch.randn_like(out_r)
    (out_r * w).sum().backward()

    xg = x.to(gpu).requires_grad_(True)
    out, (hn, cn) = mine(xg, (h0.to(gpu), c0.to(gpu)), lt)
    assert out.shape == out_r.shape
    assert (out.detach().cpu().double() - out_r.detach()).abs().max().item() < 2e-5
    assert (hn.cpu().double() - hn_r.detach()).abs().max().item() < 2e-5
    assert (cn.cpu().double() - cn_r.detach()).abs().max().item() < 2e-5
    (out * w.float().to(gpu)).sum().backward()
    gx = xg.grad.cpu().double()
    assert (gx - xr.grad).abs().max().item() < 1e-4 * max(1.0, xr.grad.abs().max().item())
    for (n, pr), (_, pm) in zip(ref.named_parameters(), mine.named_parameters()):
        err = (pm.grad.cpu().double() - pr.grad).abs().max().item()
        assert err < 1e-4 * max(1.0, pr.grad.abs().max().item()), (n, err)


def test_lstm_inference_mode_and_zero_padding(gpu):
    from idiaptts_amd.nn import LSTM
    torch.manual_seed(1)
    m = LSTM(10, 16, 1, bidirectional=True).to(gpu)
    x = torch.randn(6, 3, 10, device=gpu)
    with torch.no_grad():
        out, _ = m(x, None, torch.tensor([6, 2, 4]))
    assert (out[2:, 1] == 0).all() and (out[4:, 2] == 0).all()
    assert (out[:2, 1] != 0).any()


@pytest.mark.parametrize("cell", ["LSTM", "GRU"])
def test_trainable_initial_states_match_torch(gpu, cell):
    """RNNWrapper `train_hidden_init` (rnn_dyn/RNNWrapper.py:58-84): h_0 / c_0 are parameters of
    shape [layers*dirs, 1, H], expanded over the batch.  Output and the gradients of the initial
    states (and of the weights) must match torch.nn.LSTM / GRU on a PackedSequence fed with the
    same expanded states (torch CPU float64)."""
    from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence
    from idiaptts_amd.src.neural_networks.pytorch.models.rnn_dyn.Config import Config
    from idiaptts_amd.src.neural_networks.pytorch.models.rnn_dyn.RNNDyn import RNNWrapper
    torch.manual_seed(11)
    in_dim, H, layers, B, T = 7, 16, 2, 5, 12
    lengths = torch.tensor([12, 4, 9, 1, 12])
    lc = Config.LayerConfig(layer_type=cell, out_dim=H, num_layers=layers, bidirectional=True,
                            train_hidden_init=True, hidden_init_value=0.1)
    mine = RNNWrapper(in_dim, lc, batch_first=False, enforce_sorted=False).to(gpu)
    assert isinstance(mine.h_0, torch.nn.Parameter) and mine.h_0.shape == (layers * 2, 1, H)
    with torch.no_grad():
        mine.h_0.copy_(torch.randn_like(mine.h_0) * 0.5)
        mine.c_0.copy_(torch.randn_like(mine.c_0) * 0.5)
    ref = getattr(torch.nn, cell)(in_dim, H, layers, bidirectional=True).double()
    ref.load_state_dict({k: v.detach().cpu().double() for k, v in mine.module.state_dict().items()})
    h0 = mine.h_0.detach().cpu().double().requires_grad_(True)
    c0 = mine.c_0.detach().cpu().double().requires_grad_(True)
    x = torch.randn(T, B, in_dim)
    w = torch.randn(T, B, 2 * H)
    # reference
    hx = h0.expand(-1, B, -1).contiguous()
    hx = (hx, c0.expand(-1, B, -1).contiguous()) if cell == "LSTM" else hx
    out_ref, _ = ref(pack_padded_sequence(x.double(), lengths, enforce_sorted=False), hx)
    out_ref, _ = pad_packed_sequence(out_ref, total_length=T)
    (out_ref * w.double()).sum().backward()
    # ours
    mine.init_hidden(B)
    out, kw = mine(x.to(gpu), seq_lengths_input=lengths, max_length_inputs=T)
    (out * w.to(gpu)).sum().backward()
    assert (out.detach().cpu().double() - out_ref.detach()).abs().max() < 2e-5
    assert (mine.h_0.grad.cpu().double() - h0.grad).abs().max() < 1e-4 * max(1.0, float(h0.grad.abs().max()))
    if cell == "LSTM":
        assert (mine.c_0.grad.cpu().double() - c0.grad).abs().max() < 1e-4 * max(1.0, float(c0.grad.abs().max()))
    else:
        assert mine.c_0.grad is None or float(mine.c_0.grad.abs().max()) == 0.0
    for (n, pr), (_, pm) in zip(ref.named_parameters(), mine.module.named_parameters()):
        err = (pm.grad.cpu().double() - pr.grad).abs().max().item()
        assert err < 1e-4 * max(1.0, pr.grad.abs().max().item()), (n, err)


@pytest.mark.parametrize("cell,H", [("LSTM", 5), ("GRU", 20), ("LSTM", 100)])
def test_hidden_sizes_that_are_not_multiples_of_16(gpu, cell, H):
    """Any hidden size works (the kernels tile 16 units; other sizes run zero-padded, which is
    exact): output, final states and gradients against torch CPU float64."""
    from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence
    from idiaptts_amd import nn as inn
    torch.manual_seed(2)
    in_dim, B, T = 9, 4, 10
    lengths = torch.tensor([10, 3, 7, 10])
    mine = getattr(inn, cell)(in_dim, H, 2, bidirectional=True).to(gpu)
    ref = getattr(torch.nn, cell)(in_dim, H, 2, bidirectional=True).double()
    ref.load_state_dict({k: v.detach().cpu().double() for k, v in mine.state_dict().items()})
    x = torch.randn(T, B, in_dim)
    w = torch.randn(T, B, 2 * H)
    out_ref, hn_ref = ref(pack_padded_sequence(x.double(), lengths, enforce_sorted=False))
    out_ref, _ = pad_packed_sequence(out_ref, total_length=T)
    (out_ref * w.double()).sum().backward()
    out, hn = mine(x.to(gpu), None, lengths)
    (out * w.to(gpu)).sum().backward()
    assert out.shape == (T, B, 2 * H)
    assert (out.detach().cpu().double() - out_ref.detach()).abs().max() < 2e-5
    h_mine = hn[0] if cell == "LSTM" else hn
    h_ref = hn_ref[0] if cell == "LSTM" else hn_ref
    assert (h_mine.detach().cpu().double() - h_ref.detach()).abs().max() < 2e-5
    for (n, pr), (_, pm) in zip(ref.named_parameters(), mine.named_parameters()):
        assert pm.grad.shape == pr.grad.shape
        err = (pm.grad.cpu().double() - pr.grad).abs().max().item()
        assert err < 1e-4 * max(1.0, pr.grad.abs().max().item()), (n, err)
