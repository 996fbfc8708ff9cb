"""GPU parity of the fused optimiser tail (itts_grad_norm_accum + itts_adam_step_fused behind
HipAdam's flat arena): gradient clipping by norm / by value, Adam, parameter EMA -- against the
torch utilities the reference's handler calls (ModularModelHandlerPyTorch.py:810-831,
ExponentialMovingAverage.py:32-45)."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(gpu, seed=0):
    torch.manual_seed(seed)
    return torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Tanh(), torch.nn.Linear(16, 4),
                               torch.nn.Tanh(), torch.nn.Linear(4, 3, bias=False)).to(gpu)


def _batches(gpu, n, seed=1):
    g = torch.Generator().manual_seed(seed)
    return [(torch.randn(32, 8, generator=g).to(gpu), 3 * torch.randn(32, 3, generator=g).to(gpu))
            for _ in range(n)]


def _reference_run(model, batches, norm_type, max_norm, clip_value, decay, wd=0.0):
    opt = torch.optim.Adam(model.parameters(), lr=3e-3, weight_decay=wd)
    shadow = [p.detach().clone() for p in model.parameters()]
    for x, y in batches:
        opt.zero_grad()
        ((model(x) - y) ** 2).mean().backward()
        if norm_type is not None:
            torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm, norm_type)
        if clip_value is not None:
            torch.nn.utils.clip_grad_value_(model.parameters(), clip_value)
        opt.step()
        for s, p in zip(shadow, model.parameters()):
            s.sub_((1.0 - decay) * (s - p.detach()))
    return shadow, opt


@pytest.mark.parametrize("norm_type,max_norm,clip_value,wd", [
    (2, 0.3, None, 0.0), (float("inf"), 0.05, None, 0.0), (None, None, 0.02, 0.0),
    (2.0, 0.5, 0.03, 0.01), (None, None, None, 0.0), (2, 1e6, None, 0.0)])
def test_fused_step_matches_torch_clip_adam_ema(gpu, norm_type, max_norm, clip_value, wd):
    from idiaptts_amd.src.neural_networks.pytorch.ModularModelHandlerPyTorch import (
        ExponentialMovingAverage, HipAdam)
    batches = _batches(gpu, 6)
    ref = _model(gpu)
    mine = copy.deepcopy(ref)
    shadow_ref, opt_ref = _reference_run(ref, batches, norm_type, max_norm, clip_value, 0.9, wd)
    opt = HipAdam(mine.parameters(), lr=3e-3, weight_decay=wd)
    assert opt._arenas is not None
    ema = ExponentialMovingAverage(mine, 0.9)
    assert opt.configure_clipping(norm_type, max_norm, clip_value)
    assert opt.attach_ema(ema) and ema.fused
    for x, y in batches:
        opt.zero_grad()
        ((mine(x) - y) ** 2).mean().backward()
        opt.step()
        ema.update_params(mine)                      # no-op once fused
    for p, q in zip(mine.parameters(), ref.parameters()):
        assert torch.allclose(p, q, rtol=2e-5, atol=2e-6)
    for (n, s), q in zip(ema.model.named_parameters(), shadow_ref):
        assert torch.allclose(s, q, rtol=2e-5, atol=2e-6), n
    # the state dict is torch.optim.Adam's
    sd, sd_ref = opt.state_dict(), opt_ref.state_dict()
    assert sd["state"].keys() == sd_ref["state"].keys()
    for k in sd["state"]:
        assert int(sd["state"][k]["step"]) == int(sd_ref["state"][k]["step"]) == 6
        assert torch.allclose(sd["state"][k]["exp_avg"], sd_ref["state"][k]["exp_avg"],
                              rtol=2e-5, atol=1e-7)
        assert torch.allclose(sd["state"][k]["exp_avg_sq"], sd_ref["state"][k]["exp_avg_sq"],
                              rtol=2e-5, atol=1e-9)


def test_unsupported_norm_type_is_left_to_the_caller(gpu):
    from idiaptts_amd.src.neural_networks.pytorch.ModularModelHandlerPyTorch import HipAdam
    opt = HipAdam(_model(gpu).parameters())
    assert not opt.configure_clipping(3, 1.0, None)
    assert opt.configure_clipping(None, None, None)
    assert not HipAdam(_model(gpu).parameters(), flat=False).configure_clipping(2, 1.0, None)


def test_arena_survives_state_dict_round_trip_and_late_cuda(gpu):
    from idiaptts_amd.src.neural_networks.pytorch.ModularModelHandlerPyTorch import HipAdam
    batches = _batches(gpu, 5)
    a = _model(gpu)
    opt_a = HipAdam(a.parameters(), lr=3e-3)
    for x, y in batches[:3]:
        opt_a.zero_grad()
        ((a(x) - y) ** 2).mean().backward()
        opt_a.step()
    # resume in a fresh model / optimiser created on the CPU and moved afterwards
    b = _model("cpu", seed=5)
    opt_b = HipAdam(b.parameters(), lr=3e-3)
    assert opt_b._arenas is None
    b.to(gpu)
    b.load_state_dict(copy.deepcopy(a.state_dict()))
    opt_b.load_state_dict(copy.deepcopy(opt_a.state_dict()))
    for x, y in batches[3:]:
        for m, o in ((a, opt_a), (b, opt_b)):
            o.zero_grad()
            ((m(x) - y) ** 2).mean().backward()
            o.step()
    assert opt_b._arenas is not None
    for p, q in zip(a.parameters(), b.parameters()):
        assert torch.equal(p, q)
    assert int(opt_b.state_dict()["state"][0]["step"]) == 5


def test_grad_norm_accum_large_buffers(gpu):
    from idiaptts_amd import ops
    torch.manual_seed(3)
    x = torch.randn(3_000_001, device=gpu)
    y = torch.randn(777, device=gpu) * 5
    acc = torch.full((1,), 123.0, device=gpu)
    ops.grad_norm_accum(x, acc, 2, accumulate=False)
    ops.grad_norm_accum(y, acc, 2, accumulate=True)
    want = float((x.double() ** 2).sum() + (y.double() ** 2).sum())
    assert abs(float(acc) - want) / want < 1e-6
    ops.grad_norm_accum(x, acc, 0, accumulate=False)
    ops.grad_norm_accum(y, acc, 0, accumulate=True)
    assert float(acc) == float(torch.maximum(x.abs().max(), y.abs().max()))
    ops.grad_norm_accum(x[:0], acc, 2, accumulate=False)          # empty buffer -> 0
    assert float(acc) == 0.0


def test_ema_keeps_averaging_after_the_optimiser_is_replaced(gpu, tmp_path):
    """Loading a checkpoint (or resetting the optimiser) builds a new HipAdam with new flat
    buffers.  An existing EMA must follow: re-attached to the new optimiser (fused update) and its
    shadow restarted from the loaded parameters -- it must never keep pointing at the discarded
    optimiser, where its update would silently stop."""
    import types
    import numpy as np
    from idiaptts_amd.src.ExtendedHParams import ExtendedHParams
    from idiaptts_amd.src.neural_networks.pytorch.ModularModelHandlerPyTorch import (
        ExponentialMovingAverage, ModularModelHandlerPyTorch as Handler)
    from idiaptts_amd.src.neural_networks.pytorch.loss.NamedLoss import NamedLoss
    from idiaptts_amd.src.neural_networks.pytorch.models import rnn_dyn
    from idiaptts_amd.src.neural_networks.pytorch.models.NamedForwardWrapper import \
        NamedForwardWrapper
    torch.manual_seed(3)
    hp = types.SimpleNamespace(model_type="RNNDYN-1_TANH_16-1_FC_4", batch_first=False, dropout=0.0)
    h = Handler()
    h.create_model(NamedForwardWrapper.Config(rnn_dyn.convert_legacy_to_config((6,), hp),
                                              input_names=["questions"], batch_first=False,
                                              name="AM", output_names=["pred"]))
    hparams = ExtendedHParams.create_hparams()
    hparams.use_gpu = True
    hparams.optimiser_args["lr"] = 1e-2
    h.set_optimiser(hparams)
    h.set_losses([NamedLoss.Config(name="MSELoss_y", type_="MSELoss", seq_mask="y_mask",
                                   input_names=["y", "pred"], batch_first=False)])
    h.ema = ExponentialMovingAverage(h.model, 0.5)
    assert h.optimiser.attach_ema(h.ema) and h.ema.fused
    rng = np.random.default_rng(0)
    batch = [{"questions": rng.normal(size=(t, 6)).astype(np.float32),
              "y": rng.normal(size=(t, 4)).astype(np.float32)} for t in (9, 5)]
    data, lengths = Handler.prepare_batch(batch, batch_first=False, mask_keys=("y",))

    def shadow():
        return torch.cat([s.reshape(-1) for s in h.ema.shadow.values()]).clone()

    def params():
        return torch.cat([p.detach().reshape(-1) for p in h.model.parameters()]).clone()

    h.process_batch(data, lengths, 0, training=True)
    h.save_checkpoint(str(tmp_path), epoch=1, step=1)
    h.process_batch(data, lengths, 1, training=True)
    old_opt = h.optimiser
    h.load_checkpoint(hparams, str(tmp_path), epoch=1)
    assert h.optimiser is not old_opt
    # the checkpoint holds the averaged parameters; the shadow restarts from what was loaded
    assert torch.equal(shadow(), params())
    s0 = shadow()
    for step in range(3):
        h.process_batch(data, lengths, step + 2, training=True)
        s1 = shadow()
        assert not torch.equal(s0, s1), "the EMA stopped following the parameters"
        # shadow = 0.5 * shadow + 0.5 * param after every step
        s0 = s1
    p = params()
    assert float((s1 - p).abs().max()) > 0 and float((s1 - p).abs().max()) < 0.1
    # same with an optimiser that cannot fuse the update: falls back to the separate kernel
    h.set_optimiser("SGD", lr=1e-2)
    assert not h.ema.fused
    h.process_batch(data, lengths, 9, training=True)
    assert not torch.equal(shadow(), s1)
