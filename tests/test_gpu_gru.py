"""GPU parity of the (Bi)GRU path against torch.nn.GRU on the CPU (float64) fed with
pack_padded_sequence(enforce_sorted=False) -- the calls of rnn_dyn/RNNWrapper.py:89-102 -- and of
the fused SGD / EMA kernels against torch."""
import numpy as np
import pytest
import torch
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("in_dim,H,layers,bidir,lengths,batch_first", [
    (20, 16, 1, False, [7], False),
    (20, 16, 1, True, [9, 4, 6], False),
    (37, 32, 2, True, [12, 12, 1, 5, 9], True),
    (425, 64, 3, True, [40, 33, 17], False),
    (48, 128, 1, True, list(range(1, 20)), False),
])
def test_gru_forward_backward_match_torch(gpu, in_dim, H, layers, bidir, lengths, batch_first):
    from idiaptts_amd.nn import GRU
    torch.manual_seed(0)
    ref = torch.nn.GRU(in_dim, H, layers, bidirectional=bidir, batch_first=batch_first).double()
    mine = GRU(in_dim, H, layers, bidirectional=bidir, batch_first=batch_first)
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
    xr = x.double().requires_grad_(True)
    packed = pack_padded_sequence(xr, lt, batch_first=batch_first, enforce_sorted=False)
    out_p, hn_r = ref(packed, h0.double())
    out_r, _ = pad_packed_sequence(out_p, batch_first=batch_first, total_length=T)
    w = torch.randn_like(out_r)
    (out_r * w).sum().backward()

    xg = x.to(gpu).requires_grad_(True)
    out, hn = mine(xg, h0.to(gpu), lt)
    assert out.shape == out_r.shape
    assert (out.detach().cpu().double() - out_r.detach()).abs().max().item() < 2e-5
    assert (hn.cpu().double() - hn_r.detach()).abs().max().item() < 2e-5
    (out * w.float().to(gpu)).sum().backward()
    gx = xg.grad.cpu().double()
    assert (gx - xr.grad).abs().max().item() < 1e-4 * max(1.0, xr.grad.abs().max().item())
    for (n, pr), (_, pm) in zip(ref.named_parameters(), mine.named_parameters()):
        err = (pm.grad.cpu().double() - pr.grad).abs().max().item()
        assert err < 1e-4 * max(1.0, pr.grad.abs().max().item()), (n, err)


def test_gru_group_in_rnndyn(gpu):
    """'RNNDYN-1_TANH_32-2_BiGRU_16-1_FC_5' builds, keeps torch's state-dict keys, zero-pads."""
    import types
    from idiaptts_amd.src.neural_networks.pytorch.models import rnn_dyn
    hp = types.SimpleNamespace(model_type="RNNDYN-1_TANH_32-2_BiGRU_16-1_FC_5", batch_first=False,
                               dropout=0.0)
    torch.manual_seed(3)
    model = rnn_dyn.convert_legacy_to_config((12,), hp).create_model().to(gpu)
    keys = list(model.state_dict().keys())
    assert "2.module.weight_ih_l1_reverse" in keys and "2.h_0" in keys and "2.c_0" in keys
    model.init_hidden(3)
    x = torch.randn(9, 3, 12, device=gpu)
    lengths = torch.tensor([9, 3, 6])
    out, kw = model(x, seq_lengths_input=lengths, max_length_inputs=9)
    assert out.shape == (9, 3, 5) and torch.isfinite(out).all()
    assert torch.is_tensor(kw["hidden"]) and kw["hidden"].shape == (4, 3, 16)
    out.sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())


@pytest.mark.parametrize("momentum,dampening,nesterov,wd", [
    (0.0, 0.0, False, 0.0), (0.9, 0.0, False, 0.0), (0.9, 0.1, False, 1e-2), (0.8, 0.0, True, 1e-3)])
def test_sgd_kernel_matches_torch(gpu, momentum, dampening, nesterov, wd):
    from idiaptts_amd.src.neural_networks.pytorch.ModularModelHandlerPyTorch import HipSGD
    torch.manual_seed(0)
    p_ref = torch.nn.Parameter(torch.randn(1000, 37))
    p_hip = torch.nn.Parameter(p_ref.detach().clone().to(gpu))
    kw = dict(lr=0.05, momentum=momentum, dampening=dampening, nesterov=nesterov, weight_decay=wd)
    o_ref, o_hip = torch.optim.SGD([p_ref], **kw), HipSGD([p_hip], **kw)
    for step in range(4):
        g = torch.randn_like(p_ref)
        p_ref.grad, p_hip.grad = g.clone(), g.to(gpu)
        o_ref.step()
        o_hip.step()
        assert (p_hip.detach().cpu() - p_ref.detach()).abs().max().item() < 1e-6


def test_ema_kernel_matches_reference_formula(gpu):
    from idiaptts_amd.src.neural_networks.pytorch.ModularModelHandlerPyTorch import \
        ExponentialMovingAverage
    torch.manual_seed(0)
    model = torch.nn.Linear(19, 7).to(gpu)
    ema = ExponentialMovingAverage(model, 0.9)
    shadow = {k: v.clone().cpu().double() for k, v in ema.shadow.items()}
    for _ in range(3):
        with torch.no_grad():
            for p in model.parameters():
                p.add_(torch.randn_like(p))
        ema.update_params(model)
        for n, p in model.named_parameters():
            shadow[n] -= (1.0 - 0.9) * (shadow[n] - p.detach().cpu().double())
    for n in shadow:
        assert (ema.shadow[n].cpu().double() - shadow[n]).abs().max().item() < 1e-6
    assert all(not p.requires_grad for p in ema.model.parameters())


@pytest.mark.parametrize("nonlin,bidir,layers,batch_first", [
    ("tanh", True, 2, False), ("relu", False, 1, True), ("tanh", False, 1, False)])
def test_vanilla_rnn_matches_torch(gpu, nonlin, bidir, layers, batch_first):
    """torch.nn.RNN on a PackedSequence (RNNWrapper 'RNN' groups, legacy 'RNNTANH' / 'RNNRELU'):
    output, final states and all gradients against torch CPU float64."""
    from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence
    from idiaptts_amd.nn import RNN
    torch.manual_seed(5)
    in_dim, H, B, T = 6, 16, 4, 11
    lengths = torch.tensor([7, 11, 1, 7])
    mine = RNN(in_dim, H, layers, nonlinearity=nonlin, bidirectional=bidir,
               batch_first=batch_first).to(gpu)
    ref = torch.nn.RNN(in_dim, H, layers, nonlinearity=nonlin, bidirectional=bidir,
                       batch_first=batch_first).double()
    ref.load_state_dict({k: v.detach().cpu().double() for k, v in mine.state_dict().items()})
    shape = (B, T, in_dim) if batch_first else (T, B, in_dim)
    x = torch.randn(shape)
    ndir = 2 if bidir else 1
    w = torch.randn(shape[0], shape[1], ndir * H)
    h0 = torch.randn(layers * ndir, 1, H) * 0.3
    out_ref, hn_ref = ref(pack_padded_sequence(x.double(), lengths, batch_first=batch_first,
                                               enforce_sorted=False),
                          h0.double().expand(-1, B, -1).contiguous())
    out_ref, _ = pad_packed_sequence(out_ref, batch_first=batch_first, total_length=T)
    (out_ref * w.double()).sum().backward()
    out, hn = mine(x.to(gpu), h0.to(gpu).expand(-1, B, -1).contiguous(), lengths)
    (out * w.to(gpu)).sum().backward()
    assert (out.detach().cpu().double() - out_ref.detach()).abs().max() < 2e-5
    assert (hn.detach().cpu().double() - hn_ref.detach()).abs().max() < 2e-5
    for (n, pr), (_, pm) in zip(ref.named_parameters(), mine.named_parameters()):
        err = (pm.grad.cpu().double() - pr.grad).abs().max().item()
        assert err < 1e-4 * max(1.0, pr.grad.abs().max().item()), (n, err)


def test_rnntanh_group_in_rnndyn(gpu):
    import types
    from idiaptts_amd.src.neural_networks.pytorch.models import rnn_dyn
    hp = types.SimpleNamespace(model_type="RNNDYN-1_RELU_16-1_BiRNNTANH_8-1_FC_3", batch_first=True,
                               dropout=0.0)
    model = rnn_dyn.convert_legacy_to_config((5,), hp).create_model().to(gpu)
    assert "2.module.weight_hh_l0_reverse" in model.state_dict()
    model.init_hidden(2)
    out, kw = model(torch.randn(2, 6, 5, device=gpu), seq_lengths_input=torch.tensor([6, 2]),
                    max_length_inputs=6)
    assert out.shape == (2, 6, 3) and torch.isfinite(out).all() and kw["hidden"].shape == (2, 2, 8)
    out.sum().backward()
    assert all(p.grad is not None for p in model.parameters())
