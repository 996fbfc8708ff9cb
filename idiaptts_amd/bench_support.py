"""Synthetic workloads, the torch-CPU restatement of the reference's
acoustic-model step (the CPU baseline / checker for the NN path) and the smoke check."""
import math

import numpy as np
import torch


def utterance_lengths(n_utts, seed):
    """duration ~ U(2 s, 10 s) -> T = round(200 * duration) frames of 5 ms (LJSpeech-like)."""
    rng = np.random.default_rng(1234 + seed)
    return np.rint(200.0 * rng.uniform(2.0, 10.0, size=n_utts)).astype(np.int64)


def make_ff_batch(n_utts, seed, in_dim=425, out_dim=187, device="cpu"):
    """Packed valid frames of one mini-batch: x ~ min-max-normalised question vectors
    (in_dim-9 binary columns at 5 % density + 9 continuous), y ~ N(0,1) (mean-var-normalised cmp).
    """
    lengths = utterance_lengths(n_utts, seed)
    M = int(lengths.sum())
    g = torch.Generator().manual_seed(seed)
    x = (torch.rand(M, in_dim, generator=g) < 0.05).float()
    x[:, in_dim - 9:] = torch.rand(M, 9, generator=g)
    y = torch.randn(M, out_dim, generator=g)
    return x.to(device), y.to(device), lengths


class TorchRefFF(torch.nn.Module):
    """The module stack the reference builds for `RNNDYN-2_TANH_512-1_FC_187`
    (rnn_dyn/FFWrapper.py:63-73: nn.Sequential(Linear, Tanh) per layer, last group linear)."""

    def __init__(self, layers, acts):
        super().__init__()
        mods = []
        for (w, b), a in zip(layers, acts):
            lin = torch.nn.Linear(w.shape[1], w.shape[0])
            with torch.no_grad():
                lin.weight.copy_(w)
                lin.bias.copy_(b)
            mods.append(lin)
            if a == "tanh":
                mods.append(torch.nn.Tanh())
            elif a == "relu":
                mods.append(torch.nn.ReLU())
        self.net = torch.nn.Sequential(*mods)

    def forward(self, x):
        return self.net(x)


def torch_ref_step(model, opt, x_pad, y_pad, lengths):
    """One reference training step on a PADDED batch [B, T, D] (batch_first) exactly as
    process_dataloader does it: forward, MSELoss(none)*mask, mean_per_frame, backward, Adam."""
    B, T, _ = x_pad.shape
    mask = (torch.arange(T)[None, :] < lengths[:, None]).unsqueeze(-1).float()
    pred = model(x_pad)
    v = torch.nn.functional.mse_loss(y_pad, pred, reduction="none") * mask
    loss = (v.sum(dim=(0, 1)) / lengths.sum().float()).mean()
    opt.zero_grad()
    loss.backward()
    opt.step()
    return loss.detach()


def pad_batch(x, lengths):
    """packed [sum T, D] -> padded [B, Tmax, D] (pad_sequence, batch_first)."""
    parts = torch.split(x, [int(l) for l in lengths])
    return torch.nn.utils.rnn.pad_sequence(parts, batch_first=True)


def ff_smoke(dev):
    from .native_ff import FlatFFModel
    dims, acts = (425, 512, 512, 187), ("tanh", "tanh", None)
    model = FlatFFModel(dims, acts, device=dev, seed=1)
    ref = TorchRefFF(model.layers_cpu(), acts) if hasattr(model, "layers_cpu") else \
        TorchRefFF([(w.cpu(), b.cpu()) for w, b in model.layers()], acts)
    opt = torch.optim.Adam(ref.parameters(), lr=1e-3)
    x, y, lengths = make_ff_batch(2, seed=3)
    x, y, lengths = x[:700], y[:700], np.array([400, 300])
    lt = torch.from_numpy(lengths)
    ref_loss = torch_ref_step(ref, opt, pad_batch(x, lt), pad_batch(y, lt), lt)
    valid = torch.ones(x.shape[0], dtype=torch.uint8, device=dev)
    loss = model.train_step(x.to(dev), y.to(dev), valid, float(lengths.sum()))
    assert abs(loss.item() - ref_loss.item()) < 1e-5 * max(1.0, abs(ref_loss.item())), \
        (loss.item(), ref_loss.item())
    w_ref = ref.net[0].weight.detach()
    assert (model.weight(0).cpu() - w_ref).abs().max().item() < 1e-5
    return loss.item()


from .synthetic_audio import make_audio, make_audio_batch  # noqa: E402,F401  (torch-free module)
