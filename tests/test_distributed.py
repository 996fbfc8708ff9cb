"""N > 1 path on CPU (gloo, world_size 2): sharding, global frame count, flat-gradient
all-reduce and statistics merge reproduce the single-process result."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from idiaptts_amd import parallel
from idiaptts_amd.bench_support import TorchRefFF, make_ff_batch
from idiaptts_amd.misc.normalisation.MeanCovarianceExtractor import MeanCovarianceExtractor
from idiaptts_amd.misc.normalisation.MeanStdDevExtractor import MeanStdDevExtractor


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _layers(dims, seed):
    g = torch.Generator().manual_seed(seed)
    return [((torch.rand(n, k, generator=g) - 0.5) * 0.2, (torch.rand(n, generator=g) - 0.5) * 0.2)
            for k, n in zip(dims[:-1], dims[1:])]


def _flat_grad(model):
    return torch.cat([p.grad.reshape(-1) for p in model.parameters()])


def _local_grad(x, y, lengths, idx, n_global, dims, acts):
    """Gradient of this shard's frames with the loss normalised by the GLOBAL frame count."""
    model = TorchRefFF(_layers(dims, 7), acts).double()
    offs = np.concatenate([[0], np.cumsum(lengths)])
    rows = np.concatenate([np.arange(offs[i], offs[i + 1]) for i in idx]) if len(idx) else \
        np.zeros(0, dtype=np.int64)
    pred = model(x[rows].double())
    loss = ((pred - y[rows].double()) ** 2).sum() / (n_global * dims[-1])
    loss.backward()
    return _flat_grad(model), float(loss.detach())


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dims, acts = (12, 16, 9), ("tanh", None)
    x, y, lengths = make_ff_batch(5, seed=3, in_dim=12, out_dim=9)
    lengths = (lengths // 40).astype(np.int64) + 1
    n = int(lengths.sum())
    x, y = x[:n], y[:n]
    shards = parallel.shard_by_length(lengths, world)
    n_local = int(sum(lengths[i] for i in shards[rank]))
    n_global = parallel.global_sum(n_local)
    grad, loss = _local_grad(x, y, lengths, shards[rank], n_global, dims, acts)
    parallel.allreduce_flat_(grad)
    loss_g = parallel.global_sum(loss)
    # statistics merge
    offs = np.concatenate([[0], np.cumsum(lengths)])
    ms, mc = MeanStdDevExtractor(), MeanCovarianceExtractor()
    for i in shards[rank]:
        ms.add_sample(y[offs[i]:offs[i + 1]].double().numpy())
        mc.add_sample(y[offs[i]:offs[i + 1]].double().numpy())
    parallel.allreduce_stats_(ms)
    parallel.allreduce_stats_(mc)
    # a rank whose shard is empty (fewer utterances than ranks) still takes part in the merge
    for lone in (MeanStdDevExtractor(), MeanCovarianceExtractor()):
        if rank == 0:
            lone.add_sample(y[:7].double().numpy())
        parallel.allreduce_stats_(lone)
        if rank == 1:
            ret["lone_" + type(lone).__name__] = lone.get_params()
    if rank == 0:
        ret["n_global"] = n_global
        ret["grad"] = grad.numpy()
        ret["loss"] = loss_g
        ret["mean"], ret["std"] = ms.get_params()
        ret["cov"] = mc.get_params()[1]
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_equals_single_process():
    world = 2
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    dims, acts = (12, 16, 9), ("tanh", None)
    x, y, lengths = make_ff_batch(5, seed=3, in_dim=12, out_dim=9)
    lengths = (lengths // 40).astype(np.int64) + 1
    n = int(lengths.sum())
    x, y = x[:n], y[:n]
    assert ret["n_global"] == n
    grad, loss = _local_grad(x, y, lengths, list(range(len(lengths))), float(n), dims, acts)
    assert abs(ret["loss"] - loss) < 1e-12
    assert np.abs(ret["grad"] - grad.numpy()).max() < 1e-12
    ms, mc = MeanStdDevExtractor(), MeanCovarianceExtractor()
    ms.add_sample(y.double().numpy())
    mc.add_sample(y.double().numpy())
    assert np.allclose(ret["mean"], ms.get_params()[0], atol=1e-12)
    assert np.allclose(ret["std"], ms.get_params()[1], atol=1e-9)
    assert np.allclose(ret["cov"], mc.get_params()[1], atol=1e-9)
    for lone in (MeanStdDevExtractor(), MeanCovarianceExtractor()):
        lone.add_sample(y[:7].double().numpy())
        for got, want in zip(ret["lone_" + type(lone).__name__], lone.get_params()):
            assert np.allclose(got, want, atol=1e-12)


def test_shard_by_length_is_balanced_and_complete():
    rng = np.random.default_rng(0)
    lengths = rng.integers(400, 2000, size=64)
    for world in (1, 2, 4, 8):
        shards = parallel.shard_by_length(lengths, world)
        assert sorted(i for s in shards for i in s) == list(range(64))
        loads = [sum(int(lengths[i]) for i in s) for s in shards]
        assert max(loads) - min(loads) <= int(lengths.max())
    assert parallel.shard_by_length([5, 5], 4) == [[0], [1], [], []]


def _handler_dp_worker(rank, world, port, ret):
    """The module-stack handler's DP step on CPU stand-ins: prepare_batch(shard=...) picks this
    rank's samples of the global batch, the loss is a mean over the LOCAL frames, _dp_weight and
    allreduce_module_grads_ turn local gradients into the global-batch gradient."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from idiaptts_amd.src.neural_networks.pytorch.ModularModelHandlerPyTorch import \
        ModularModelHandlerPyTorch as Handler
    batch, model = _dp_case()
    data, lengths = Handler.prepare_batch(batch, common_divisor=world, batch_first=False,
                                          mask_keys=("y",), shard=(rank, world))
    h = Handler()
    h.model = model
    h.losses = [type("L", (), {"seq_mask": "y_mask"})()]
    w = h._dp_weight(lengths, torch.device("cpu"))
    loss = _masked_mean_loss(model, data, lengths)
    loss.backward()
    parallel.allreduce_module_grads_(list(model.parameters()), w)
    loss_g = parallel.allreduce_flat_(loss.detach() * w)
    if rank == 0:
        ret["ids"] = list(data["_id_list"])
        ret["grad"] = _flat_grad(model).numpy()
        ret["loss"] = float(loss_g)
    dist.barrier()
    dist.destroy_process_group()


def _dp_case():
    g = torch.Generator().manual_seed(11)
    lens = [7, 3, 5, 9, 2]          # 5 samples: the remainder sample is dropped for world 2
    batch = [{"x": torch.randn(t, 6, generator=g, dtype=torch.float64).numpy(),
              "y": torch.randn(t, 4, generator=g, dtype=torch.float64).numpy(),
              "_id_list": "utt%d" % i} for i, t in enumerate(lens)]
    model = torch.nn.Linear(6, 4).double()
    with torch.no_grad():
        model.weight.copy_(torch.randn(4, 6, generator=g, dtype=torch.float64) * 0.3)
        model.bias.copy_(torch.randn(4, generator=g, dtype=torch.float64) * 0.1)
    return batch, model


def _masked_mean_loss(model, data, lengths):
    pred = model(data["x"])
    se = ((pred - data["y"]) ** 2) * data["y_mask"]
    return se.sum() / (float(sum(lengths["y_mask"])) * pred.shape[-1])


def test_handler_data_parallel_step_equals_single_process():
    from idiaptts_amd.src.neural_networks.pytorch.ModularModelHandlerPyTorch import \
        ModularModelHandlerPyTorch as Handler
    world = 2
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_handler_dp_worker, args=(r, world, port, ret))
             for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    batch, model = _dp_case()
    data, lengths = Handler.prepare_batch(batch, common_divisor=world, batch_first=False,
                                          mask_keys=("y",))
    assert data["_id_list"] == ["utt0", "utt1", "utt2", "utt3"]      # remainder dropped
    assert ret["ids"] == ["utt0", "utt2"]                            # rank 0 of 2
    loss = _masked_mean_loss(model, data, lengths)
    loss.backward()
    assert abs(ret["loss"] - float(loss)) < 1e-12
    assert np.abs(ret["grad"] - _flat_grad(model).numpy()).max() < 1e-12
