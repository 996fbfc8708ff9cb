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
