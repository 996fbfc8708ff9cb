"""The N > 1 path on the one GPU of the test box: two processes share device 0 and talk over gloo
(RCCL refuses two ranks on one device; the collective calls are the same torch.distributed ones).
Synchronous data parallelism must reproduce the single-process step on the concatenated batch
(reference semantics ModularModelHandlerPyTorch.py:392-395, 732-735)."""
import json
import os
import socket
import subprocess
import sys
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIMS, ACTS = (425, 512, 512, 187), ("tanh", "tanh", None)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _init(rank, world, port):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    return torch.device("cuda", 0)


def _ff_case():
    from idiaptts_amd.bench_support import make_ff_batch
    x, y, lengths = make_ff_batch(6, seed=3)
    return x, y, np.asarray(lengths)


def _ff_steps(dev, x, y, n_global, world, steps=3):
    from idiaptts_amd.native_ff import FlatFFModel
    model = FlatFFModel(DIMS, ACTS, device=dev, seed=0)
    xd, yd = torch.as_tensor(x).to(dev), torch.as_tensor(y).to(dev)
    valid = torch.ones(xd.shape[0], dtype=torch.uint8, device=dev)
    losses, g1 = [], None
    for s in range(steps):
        loss = model.train_step(xd, yd, valid, float(n_global), lr=1e-3, world_size=world)
        losses.append(float(loss))
        if s == 0:
            g1 = model.grads.cpu().numpy().copy()
    torch.cuda.synchronize()
    return model.params.cpu().numpy(), g1, losses


def _ff_dp_worker(rank, world, port, ret):
    import torch.distributed as dist
    dev = _init(rank, world, port)
    x, y, lengths = _ff_case()
    offs = np.concatenate([[0], np.cumsum(lengths)])
    rows = np.concatenate([np.arange(offs[u], offs[u + 1]) for u in range(rank, len(lengths), world)])
    x, y = np.asarray(x)[rows], np.asarray(y)[rows]
    ret[rank] = _ff_steps(dev, x, y, int(lengths.sum()), world)
    dist.barrier()
    dist.destroy_process_group()


def _spawn(worker, world=2):
    import torch.multiprocessing as mp
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    return ret


def _assert_params_close(got, want, steps):
    """Adam turns a gradient g into a step of lr * g / (|g| + eps): entries whose gradient is at
    rounding level may step differently, everything else must agree closely."""
    d = np.abs(got - want)
    assert np.quantile(d, 0.999) < 2e-6
    assert d.max() < 2.1e-3 * steps


def test_flat_ff_train_step_world2_equals_single_process(gpu):
    ret = _spawn(_ff_dp_worker)
    x, y, lengths = _ff_case()
    params, g1, losses = _ff_steps(gpu, np.asarray(x), np.asarray(y), int(lengths.sum()), 1)
    for rank in (0, 1):
        p_r, g_r, _ = ret[rank]
        assert np.abs(g_r - g1).max() < 1e-5 * np.abs(g1).max()       # all-reduced gradient
        _assert_params_close(p_r, params, 3)
    assert np.array_equal(ret[0][0], ret[1][0])                        # ranks stay in lock step
    for s in range(3):     # local losses are shares of the global mean
        assert abs(ret[0][2][s] + ret[1][2][s] - losses[s]) < 1e-5 * max(1.0, losses[s])


def _bilstm_case():
    rng = np.random.default_rng(5)
    lens = [23, 9, 17, 30, 4, 12]
    return [{"questions": rng.normal(size=(t, 20)).astype(np.float32),
             "acoustic_features": rng.normal(size=(t, 7)).astype(np.float32)} for t in lens]


def _bilstm_handler(dev):
    from idiaptts_amd.src.neural_networks.pytorch.ModularModelHandlerPyTorch import \
        HipAdam, ModularModelHandlerPyTorch as Handler
    from idiaptts_amd.src.neural_networks.pytorch.loss.NamedLoss import NamedLoss
    from idiaptts_amd.src.neural_networks.pytorch.models import rnn_dyn
    from idiaptts_amd.src.neural_networks.pytorch.models.NamedForwardWrapper import \
        NamedForwardWrapper
    torch.manual_seed(21)
    hp = types.SimpleNamespace(model_type="RNNDYN-1_TANH_32-2_BiLSTM_32-1_FC_7", batch_first=False,
                               dropout=0.0)
    h = Handler()
    h.create_model(NamedForwardWrapper.Config(rnn_dyn.convert_legacy_to_config((20,), hp),
                                              input_names=["questions"], batch_first=False,
                                              name="AM", output_names=["pred_acoustic_features"]))
    h.set_optimiser("Adam", lr=1e-3)
    assert isinstance(h.optimiser, HipAdam)
    h.set_losses([NamedLoss.Config(name="MSELoss_acoustic_features", type_="MSELoss",
                                   seq_mask="acoustic_features_mask",
                                   input_names=["acoustic_features", "pred_acoustic_features"],
                                   batch_first=False)])
    return h, Handler


def _bilstm_steps(dev, shard, steps=3):
    h, Handler = _bilstm_handler(dev)
    data, lengths = Handler.prepare_batch(_bilstm_case(), common_divisor=2, batch_first=False,
                                          mask_keys=("acoustic_features",), shard=shard)
    losses, g1 = [], None
    for s in range(steps):
        ld, _ = h.process_batch(data, lengths, s, training=True)
        losses.append(ld["MSELoss_acoustic_features"])
        if s == 0:
            assert h.optimiser._arenas is not None                    # flat HipAdam arena in use
            g1 = torch.cat([p.grad.reshape(-1) for p in h.model.parameters()]).cpu().numpy()
    torch.cuda.synchronize()
    params = torch.cat([p.detach().reshape(-1) for p in h.model.parameters()]).cpu().numpy()
    return params, g1, losses


def _bilstm_dp_worker(rank, world, port, ret):
    import torch.distributed as dist
    dev = _init(rank, world, port)
    ret[rank] = _bilstm_steps(dev, (rank, world))
    dist.barrier()
    dist.destroy_process_group()


def test_bilstm_handler_step_world2_equals_single_process(gpu):
    ret = _spawn(_bilstm_dp_worker)
    params, g1, losses = _bilstm_steps(gpu, None)
    for rank in (0, 1):
        p_r, g_r, l_r = ret[rank]
        assert np.abs(g_r - g1).max() < 2e-5 * np.abs(g1).max()
        _assert_params_close(p_r, params, 3)
        for a, b in zip(l_r, losses):                                  # global loss on every rank
            assert abs(a - b) < 1e-5 * max(1.0, abs(b))
    assert np.array_equal(ret[0][0], ret[1][0])


def test_bench_starts_its_own_ranks(gpu):
    """`python bench.py --gpus 2` without a launcher: the parent spawns torch.distributed.run
    before touching HIP, rank 0 prints the one JSON line."""
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2",
                          "--share-gpu", "--steps", "3", "--warmup", "1", "--world-utts", "4",
                          "--bilstm-utts", "4", "--no-cpu-baseline"], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and "shared_gpu" in out and out["value"] > 0
    assert out["world"]["n_gpus"] == 2 and out["bilstm"]["n_gpus"] == 2
