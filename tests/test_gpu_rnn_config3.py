"""Parity of the recurrent kernels at the sizes BASELINE config 3 runs (3 x 512 BiLSTM / BiGRU,
in_dim 425, up to 80 padded utterances per batch): the forward instantiations with 2, 3 and 4
batch tiles, the loop over more than four tiles (B > 64), the multi-tile backward, the K loops at
H = 512 and H = 1024 -- against torch.nn.LSTM / GRU on the CPU in float64 fed with
pack_padded_sequence(enforce_sorted=False), the calls of rnn_dyn/RNNWrapper.py:89-102 -- and one
whole handler step of RNNDYN-3_BiLSTM_512-1_FC_187 against the reference's own module stack
(tests/golden/config3_step.npz, made by tests/golden/make_golden.py --config3).

Tolerances (fp32 kernels vs an fp64 reference): outputs / final states 2e-5 absolute on O(1)
values, every gradient 1e-4 relative to the largest entry of that gradient."""
import os
import sys
import types

import numpy as np
import pytest
import torch
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def _ragged_lengths(B, T, seed):
    """lengths that cross every 16-row tile boundary: a few full-length rows, a spread, and
    length-1 / length-2 rows, in shuffled (unsorted) order"""
    rng = np.random.default_rng(seed)
    lens = rng.integers(1, T + 1, size=B)
    lens[0] = T
    lens[1] = T
    lens[-1] = 1
    if B > 3:
        lens[-2] = 1
        lens[2] = 2
    return torch.from_numpy(lens[rng.permutation(B)].astype(np.int64))


def _run_pair(cell, gpu, in_dim, H, layers, B, T, seed, with_h0=False):
    from idiaptts_amd import nn as inn
    torch.manual_seed(seed)
    mine = getattr(inn, cell)(in_dim, H, layers, bidirectional=True).to(gpu)
    ref = getattr(torch.nn, cell)(in_dim, H, layers, bidirectional=True).double()
    ref.load_state_dict({k: v.detach().cpu().double() for k, v in mine.state_dict().items()})
    lens = _ragged_lengths(B, T, seed)
    x = torch.randn(T, B, in_dim)
    for b, l in enumerate(lens.tolist()):
        x[l:, b] = 3.0                       # garbage in the padding must not matter
    w = torch.randn(T, B, 2 * H)
    hx_ref = hx_mine = None
    if with_h0:
        h0 = torch.randn(layers * 2, 1, H).expand(-1, B, -1).contiguous() * 0.3
        c0 = torch.randn(layers * 2, 1, H).expand(-1, B, -1).contiguous() * 0.3
        hx_ref = (h0.double(), c0.double()) if cell == "LSTM" else h0.double()
        hx_mine = (h0.to(gpu), c0.to(gpu)) if cell == "LSTM" else h0.to(gpu)
    xr = x.double().requires_grad_(True)
    out_p, hn_ref = ref(pack_padded_sequence(xr, lens, enforce_sorted=False), hx_ref)
    out_ref, _ = pad_packed_sequence(out_p, total_length=T)
    (out_ref * w.double()).sum().backward()

    xg = x.to(gpu).requires_grad_(True)
    out, hn = mine(xg, hx_mine, lens)
    (out * w.to(gpu)).sum().backward()
    torch.cuda.synchronize()
    assert out.shape == out_ref.shape
    assert (out.detach().cpu().double() - out_ref.detach()).abs().max().item() < 2e-5
    if cell == "LSTM":
        assert (hn[0].cpu().double() - hn_ref[0].detach()).abs().max().item() < 2e-5
        assert (hn[1].cpu().double() - hn_ref[1].detach()).abs().max().item() < 2e-5
    else:
        assert (hn.cpu().double() - hn_ref.detach()).abs().max().item() < 2e-5
    gx = xg.grad.cpu().double()
    assert (gx - xr.grad).abs().max().item() < 1e-4 * max(1.0, xr.grad.abs().max().item())
    for (n, pr), (_, pm) in zip(ref.named_parameters(), mine.named_parameters()):
        err = (pm.grad.cpu().double() - pr.grad).abs().max().item()
        assert err < 1e-4 * max(1.0, pr.grad.abs().max().item()), (n, err)


@pytest.mark.parametrize("B,T", [(17, 44), (33, 40), (49, 36), (64, 40), (80, 32)])
def test_bilstm_3x512_matches_torch(gpu, B, T):
    _run_pair("LSTM", gpu, 425, 512, 3, B, T, seed=100 + B)


@pytest.mark.parametrize("B,T", [(17, 44), (33, 40), (49, 36), (64, 40), (80, 32)])
def test_bigru_3x512_matches_torch(gpu, B, T):
    _run_pair("GRU", gpu, 425, 512, 3, B, T, seed=200 + B)


@pytest.mark.parametrize("cell,B,T", [("LSTM", 17, 24), ("GRU", 64, 20), ("LSTM", 80, 12)])
def test_hidden_1024_k_loop_matches_torch(gpu, cell, B, T):
    """H = 1024: the forward K loop runs more than one chunk per wave, backward K = 4H (3H) = 4096."""
    _run_pair(cell, gpu, 96, 1024, 1, B, T, seed=300 + B, with_h0=True)


@pytest.mark.parametrize("cell", ["LSTM", "GRU"])
def test_initial_states_at_config3_size(gpu, cell):
    _run_pair(cell, gpu, 64, 512, 2, 40, 30, seed=7, with_h0=True)


def _wrapped(model_type, in_dim):
    from idiaptts_amd.src.neural_networks.pytorch.models import rnn_dyn
    from idiaptts_amd.src.neural_networks.pytorch.models.NamedForwardWrapper import \
        NamedForwardWrapper
    hp = types.SimpleNamespace(model_type=model_type, batch_first=False, dropout=0.0)
    return NamedForwardWrapper.Config(rnn_dyn.convert_legacy_to_config((in_dim,), hp),
                                      input_names=["questions"], batch_first=False, name="AM",
                                      output_names=["pred_acoustic_features"])


def test_config3_handler_step_matches_reference_stack(gpu, golden_dir):
    """RNNDYN-3_BiLSTM_512-1_FC_187 (16.6 M parameters), 33 ragged rows: prediction, loss, every
    parameter gradient (norm, sum, 48 sampled entries) and the parameters after one Adam step
    against the reference's module stack + torch.optim.Adam on the CPU."""
    import config3_data as c3
    from idiaptts_amd.src.neural_networks.pytorch.ModularModelHandlerPyTorch import \
        ModularModelHandlerPyTorch as Handler
    from idiaptts_amd.src.neural_networks.pytorch.loss.NamedLoss import NamedLoss
    g = np.load(os.path.join(golden_dir, "config3_step.npz"))
    h = Handler()
    h.create_model(_wrapped(c3.MODEL_TYPE, c3.IN_DIM))
    shapes = {k: tuple(v.shape) for k, v in h.model.state_dict().items()}
    assert sorted(shapes) == list(g["keys"])
    h.model.load_state_dict({k: torch.from_numpy(v) for k, v in c3.state(shapes).items()})
    h.set_optimiser("Adam", lr=1e-3)
    h.set_losses([NamedLoss.Config(name="MSELoss_acoustic_features", type_="MSELoss",
                                   seq_mask="acoustic_features_mask",
                                   input_names=["acoustic_features", "pred_acoustic_features"],
                                   batch_first=False)])
    x, y, lens = c3.batch()
    assert np.array_equal(lens, g["lens"])
    lens_t = torch.from_numpy(lens)
    T, B = x.shape[:2]
    data = {"questions": torch.from_numpy(x), "acoustic_features": torch.from_numpy(y),
            "acoustic_features_mask": Handler.sequence_mask(lens_t, T, batch_first=False)}
    lengths = {"questions": lens_t, "acoustic_features": lens_t, "acoustic_features_mask": lens_t}
    before = {k: p.detach().clone() for k, p in h.model.named_parameters()}
    losses, out = h.process_batch(data, lengths, step=1, training=True)
    torch.cuda.synchronize()
    loss = losses["MSELoss_acoustic_features"]
    assert abs(loss - float(g["loss"])) < 1e-5 * max(1.0, abs(float(g["loss"])))
    pred = out["pred_acoustic_features"].detach().cpu().numpy()[:, [0, B - 1]]
    assert np.abs(pred - g["pred_rows"]).max() < 2e-5
    for k, p in h.model.named_parameters():
        grad = p.grad.detach().cpu().numpy().reshape(-1).astype(np.float64)
        idx = c3.sample_index(k, grad.size)
        gn = float(g["gnorm_" + k])
        assert abs(np.sqrt((grad ** 2).sum()) - gn) < 1e-4 * gn, k
        scale = max(np.abs(g["gsamp_" + k]).max(), gn / np.sqrt(grad.size))
        assert np.abs(grad[idx] - g["gsamp_" + k]).max() < 1e-4 * scale, k
        assert abs(grad.sum() - float(g["gsum_" + k])) < 1e-4 * gn * np.sqrt(grad.size), k
        # Adam's first step moves every entry by lr * g / (|g| + eps): +-1e-3 unless g is tiny
        # (entries whose gradient is within the gradient tolerance of zero may flip direction)
        new = p.detach().cpu().numpy().reshape(-1)[idx]
        solid = np.abs(g["gsamp_" + k]) > 1e-2 * scale
        assert solid.sum() >= 8, k
        assert np.abs(new - g["psamp_" + k])[solid].max() < 2e-5, k
        assert np.abs(new - g["psamp_" + k]).max() < 2.1e-3, k
        assert not torch.equal(before[k], p.detach()), k


@pytest.mark.parametrize("cell", ["LSTM", "GRU"])
def test_persistent_recurrence_equals_the_step_kernels_and_falls_back(gpu, cell):
    """The persistent per-XCD recurrences (csrc/rnn_persist.h; taken by default at H = 512: forward
    for both cells, backward for the LSTM) against the per-step kernels on a ragged bidirectional
    batch -- outputs, final states and every gradient --, and their safety net: a launch that
    finds the abort flag raised (test hook) is redone by the step kernels, with a message, and the
    result is the same.  Child processes: a launch that gave up switches the path off for the
    rest of its process."""
    import subprocess
    import sys
    code = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, %r)
from idiaptts_amd import nn as inn
dev = torch.device("cuda", 0)
torch.manual_seed(7)
lens = torch.tensor([301, 280, 280, 150, 97, 96, 31, 30, 17, 16, 15, 5, 4, 3, 2, 2, 1, 1], dtype=torch.int64)
layer = getattr(inn, %r)(96, 512, 1, bidirectional=True).to(dev)
x = torch.randn(int(lens.max()), len(lens), 96, device=dev, requires_grad=True)
w_out = torch.randn(int(lens.max()), len(lens), 1024, device=dev) / 8
outs = []
for mode in ("0", "1"):
    os.environ["ITTS_RNN_PERSISTENT"] = mode
    o, st = layer(x, None, lens)
    st = list(st) if isinstance(st, (tuple, list)) else [st]
    loss = (o * w_out).sum()
    grads = torch.autograd.grad(loss, [x] + list(layer.parameters()))
    outs.append([o.detach()] + [q.detach() for q in st] + list(grads))
torch.cuda.synchronize()
d = max(float((a - b).abs().max() / max(1.0, float(b.abs().max()))) for a, b in zip(*outs))
assert d < 1e-5, d
print("max difference %%.2e" %% d)
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), cell)
    env = dict(os.environ)
    env.pop("ITTS_RNN_PERSIST_TEST_ABORT", None)
    res = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "gave up" not in res.stderr
    env["ITTS_RNN_PERSIST_TEST_ABORT"] = "1"
    res = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "gave up" in res.stderr


@pytest.mark.parametrize("cell,H", [("LSTM", 256), ("GRU", 384), ("LSTM", 136)])
def test_other_hidden_sizes_on_the_persistent_recurrences(gpu, cell, H, monkeypatch):
    """Hidden sizes 128..511 (the reference's RNNWrapper takes any, rnn_dyn/RNNWrapper.py:45-54) run zero-padded to
    the 512 units the persistent recurrences are built for when the batch fits their rounds (nn/modules.py,
    _persistent_width): the same outputs, states and gradients as the per-step kernels at the layer's own width --
    a padded unit never reaches a real one --, within the budget the persistent path has against the step kernels
    at 512, and every gradient has the parameter's own shape."""
    from idiaptts_amd import nn as inn
    from idiaptts_amd.nn import modules
    torch.manual_seed(11)
    lens = torch.tensor([211, 190, 190, 150, 97, 96, 31, 30, 17, 16, 15, 5, 4, 3, 2, 2, 1, 1], dtype=torch.int64)
    layer = getattr(inn, cell)(96, H, 2, bidirectional=True).to(gpu)
    x = torch.randn(int(lens.max()), len(lens), 96, device=gpu, requires_grad=True)
    w_out = torch.randn(int(lens.max()), len(lens), 2 * H, device=gpu) / 8
    assert modules._persistent_width(H, len(lens), 2, gpu) == 512          # (256 CUs: the rule applies on this box)
    assert modules._persistent_width(H, 4096, 2, gpu) is None              # too many rounds
    assert modules._persistent_width(1024, 16, 2, gpu) is None and modules._persistent_width(512, 16, 2, gpu) is None
    outs = []
    for pad in ("0", "1"):
        monkeypatch.setenv("ITTS_RNN_PAD_HIDDEN", pad)
        o, st = layer(x, None, lens)
        st = list(st) if isinstance(st, (tuple, list)) else [st]
        grads = torch.autograd.grad((o * w_out).sum(), [x] + list(layer.parameters()))
        for g, p in zip(grads[1:], layer.parameters()):
            assert g.shape == p.shape
        outs.append([o.detach()] + [q.detach() for q in st] + list(grads))
    for a, b in zip(*outs):
        assert a.shape == b.shape
        assert float((a - b).abs().max()) <= 1e-5 * max(1.0, float(b.abs().max()))
