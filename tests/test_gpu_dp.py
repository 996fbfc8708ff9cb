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
    h.optimiser.overlap_bucket_elems = 2000        # several buckets in this small model (the default is a layer's worth)
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
    if shard is not None:
        # the gradient buckets went into their all-reduce WHILE backward ran (hooks), not after it
        ov = h.optimiser.last_overlap
        assert ov is not None and ov["buckets_during_backward"] >= 3 and ov["collectives"] == ov["buckets_during_backward"], ov
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


def test_bench_with_eight_ranks_on_one_device(gpu):
    """The driver's `bench.py --gpus 8` protocol with all eight ranks on this one device (gloo collectives; never a
    measurement): rendezvous on 127.0.0.1, per-rank seeds, barrier + max-over-ranks timing, one JSON line from rank
    0, both forms of BASELINE config 3 (64 utterances per GPU, and 64 in all = 8 per GPU) and the exchange plan the
    first real SCALE record is to be checked against (reference hook: ModularModelHandlerPyTorch.py:732-735,
    :757-763)."""
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--share-gpu",
                          "--steps", "2", "--warmup", "1", "--ramp-steps", "0", "--utts-per-gpu", "2",
                          "--world-utts", "0", "--bilstm-utts", "8", "--trainer-utts", "0", "--no-cpu-baseline"],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1500)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and "shared_gpu" in out and out["value"] > 0 and out["scaling"] == "weak"
    assert out["config"]["utts_per_gpu"] == 2
    assert out["bilstm"]["n_gpus"] == 8 and out["bilstm"]["utterances_global"] == 64
    assert out["bilstm_global_batch"]["utterances_global"] == 8 and out["bilstm_global_batch"]["n_gpus"] == 8
    plan = out["exchange_plan"]
    assert plan["n_gpus"] == 8 and plan["bilstm_bigru_step"]["bytes_per_rank_per_step"] > 60e6
    assert 0 < plan["config_3_as_worded"]["predicted_speedup_at_8_gpus"]["64_utterances_in_all"] < 2.5
    assert plan["config_3_as_worded"]["predicted_speedup_at_8_gpus"]["64_utterances_per_gpu"] > 6


def _rccl_one_rank_worker(ret_path):
    """Child process: RCCL with ONE rank on cuda:0, every collective of the N > 1 path forced on."""
    import torch.distributed as dist
    from idiaptts_amd import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    x, y, _ = _ff_case()
    ref_params, ref_g1, ref_losses = _ff_steps(dev, x, y, x.shape[0], 1)     # no process group yet
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    parallel.force_collectives(True)
    assert parallel._active()
    params, g1, losses = _ff_steps(dev, x, y, x.shape[0], 1)                 # all-reduce branch taken
    # HipAdam on flat arenas: the in-place all-reduce of the arena, then a step
    from idiaptts_amd.src.neural_networks.pytorch.ModularModelHandlerPyTorch import HipAdam
    torch.manual_seed(0)
    lin = torch.nn.Linear(64, 32).to(dev)
    lin2 = torch.nn.Linear(64, 32).to(dev)
    lin2.load_state_dict(lin.state_dict())
    xin = torch.randn(16, 64, device=dev)
    outs = []
    for m, sync in ((lin, False), (lin2, True)):
        opt = HipAdam(m.parameters(), lr=1e-2)
        for _ in range(3):
            opt.zero_grad()
            m(xin).pow(2).mean().backward()
            if sync:
                assert opt.allreduce_grads_(1.0) or True
            opt.step()
        outs.append(torch.cat([p.detach().flatten() for p in m.parameters()]).cpu().numpy())
    # broadcast of mixed host / device state (torch.optim.Adam keeps `step` on the host)
    adam = torch.optim.Adam(lin.parameters(), lr=1e-3)
    lin(xin).sum().backward()
    adam.step()
    state = [v for st in adam.state.values() for v in st.values() if torch.is_tensor(v)]
    before = [t.detach().cpu().clone() for t in state]
    parallel.broadcast_tensors_(state)
    assert any(t.device.type == "cpu" for t in state)
    same_state = all(torch.equal(a, b.detach().cpu()) for a, b in zip(before, state))
    total = parallel.global_sum(3.5, device=dev)
    seed = parallel.broadcast_int(1234, device=dev)
    parallel.barrier()
    torch.cuda.synchronize()
    dist.destroy_process_group()
    with open(ret_path, "w") as f:
        json.dump({"ff_params_equal": bool(np.array_equal(params, ref_params)),
                   "ff_grads_equal": bool(np.array_equal(g1, ref_g1)),
                   "ff_losses_equal": losses == ref_losses,
                   "hipadam_equal": bool(np.array_equal(outs[0], outs[1])),
                   "state_unchanged": bool(same_state), "global_sum": total, "seed": seed}, f)


def test_rccl_with_one_rank_takes_every_collective_branch(gpu, tmp_path):
    """RCCL proven as far as one GPU allows: init_process_group('nccl', world_size=1), then the flat
    FF step with its per-layer asynchronous all-reduces, HipAdam.allreduce_grads_, the staged
    broadcast of mixed host / device optimiser state, global_sum / broadcast_int / barrier and
    destroy_process_group -- results bit-equal to the run without a process group (a one-rank sum
    is the identity).  Runs in a fresh interpreter: the communicator must not leak into the other
    tests of this process."""
    ret = os.path.join(str(tmp_path), "ret.json")
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); "
            "import test_gpu_dp as t; t._rccl_one_rank_worker(%r)" % (ROOT, os.path.join(ROOT, "tests"), ret))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    res = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:]
    out = json.load(open(ret))
    assert out["ff_params_equal"] and out["ff_grads_equal"] and out["ff_losses_equal"], out
    assert out["hipadam_equal"] and out["state_unchanged"], out
    assert out["global_sum"] == 3.5 and out["seed"] == 1234


def _native_comm_worker(ret_path):
    from idiaptts_amd import lib, parallel
    lib.require_gpu()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    comm = parallel.NativeComm(rank=0, world=1)
    g = torch.Generator(device=dev).manual_seed(0)
    grads = torch.randn(581_307, generator=g, device=dev)          # the FF model's flat gradient arena
    stats = torch.randn(1 + 187 + 187 * 187, generator=g, device=dev, dtype=torch.float64)
    want_g, want_s = grads.clone(), stats.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    comm.allreduce_flat_(grads)                                    # current stream
    comm.allreduce_flat_(stats, op="sum", stream=side)             # a stream of the caller's
    comm.allreduce_flat_(grads, op="avg")
    comm.allreduce_flat_(grads, op="max")
    comm.allreduce_flat_(grads[:0])                                # empty: no-op
    torch.cuda.synchronize()
    ok_g, ok_s = bool(torch.equal(grads, want_g)), bool(torch.equal(stats, want_s))
    bad = None
    try:
        lib.check(lib.load().itts_allreduce_flat(grads.data_ptr(), 4, 5, 0, comm.comm, None), "x")
    except lib.IttsError as e:
        bad = str(e)
    comm.close()
    with open(ret_path, "w") as f:
        json.dump({"grads_equal": ok_g, "stats_equal": ok_s, "bad_dtype": bad}, f)


def test_allreduce_flat_export_with_a_one_rank_communicator(gpu, tmp_path):
    """SURVEY.md section 8(b) `allreduce_flat`: the library's own RCCL communicator (unique id, init,
    in-place all-reduce of float32 / float64 buffers on the current and on a caller's stream, every
    reduction, destroy).  A one-rank reduction is the identity -- what one GPU can prove is that RCCL
    is found, the communicator works on the caller's streams and the buffers come back unchanged."""
    ret = os.path.join(str(tmp_path), "ret.json")
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); "
            "import test_gpu_dp as t; t._native_comm_worker(%r)" % (ROOT, os.path.join(ROOT, "tests"), ret))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    res = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:]
    out = json.load(open(ret))
    assert out["grads_equal"] and out["stats_equal"], out
    assert out["bad_dtype"] and "dtype" in out["bad_dtype"], out


def test_bench_counts_devices_without_touching_hip():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    n = bench.visible_gpus()
    assert n == torch.cuda.device_count()


def _native_comm_two_rank_worker(rank, port, ret_dir):
    """One of two processes, each on its own GPU: the library's communicator against torch.distributed's."""
    import torch.distributed as dist
    from idiaptts_amd import lib, parallel
    lib.require_gpu()
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=rank, world_size=2, device_id=dev)
    out = {"version": int(lib.load().itts_comm_version())}
    with parallel.NativeComm(rank=rank, world=2) as comm:         # id broadcast over the process group's store
        g = torch.Generator(device=dev).manual_seed(10 + rank)
        for name, dtype, n in (("f32", torch.float32, 581_307), ("f64", torch.float64, 1 + 187 + 187 * 187)):
            mine = torch.randn(n, generator=g, device=dev, dtype=dtype)
            for op, ref_op in (("sum", dist.ReduceOp.SUM), ("avg", dist.ReduceOp.AVG), ("max", dist.ReduceOp.MAX)):
                a, b = mine.clone(), mine.clone()
                comm.allreduce_flat_(a, op=op)
                dist.all_reduce(b, op=ref_op)
                torch.cuda.synchronize()
                out["{}_{}".format(name, op)] = bool(torch.equal(a, b)) and not bool(torch.equal(a, mine))
    dist.barrier()
    dist.destroy_process_group()
    with open(os.path.join(ret_dir, "rank{}.json".format(rank)), "w") as f:
        json.dump(out, f)


def test_native_comm_two_ranks_equal_torch_distributed(gpu, tmp_path):
    """a one-rank reduction is the identity and proves nothing about the id broadcast, the
    multi-rank ncclCommInitRank, or the enum values the binding hard-codes.  Two processes on two GPUs: sum,
    average and maximum of float32 / float64 buffers through itts_allreduce_flat equal torch.distributed's
    bit for bit.  Skipped where fewer than two GPUs are visible (the one-GPU boxes of the test pool)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = []
    for rank in range(2):
        code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import test_gpu_dp as t; "
                "t._native_comm_two_rank_worker(%d, %d, %r)" % (ROOT, os.path.join(ROOT, "tests"), rank, port, str(tmp_path)))
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    for p in procs:
        out, _ = p.communicate(timeout=600)
        assert p.returncode == 0, out[-3000:]
    for rank in range(2):
        res = json.load(open(os.path.join(str(tmp_path), "rank{}.json".format(rank))))
        assert res.pop("version") >= 21000
        assert all(res.values()), res


def test_rccl_version_is_one_the_binding_knows(gpu):
    from idiaptts_amd import lib
    v = int(lib.load().itts_comm_version())
    assert 21000 <= v < 30000, v


def _trainer_dp_worker(rank, world, port, ret, root, cache):
    """Child: AcousticModelTrainer.train on the reference's fixture, two ranks sharing device 0 over gloo."""
    import faulthandler
    import logging
    import torch.distributed as dist
    faulthandler.dump_traceback_later(int(os.environ.get("ITTS_TEST_HANG_S", "150")), exit=True)   # a rank stuck in a collective says where, and ends
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    _init(rank, world, port)
    os.environ["ITTS_RNN_PERSISTENT"] = "0"
    from fixture_dirs import materialise
    from idiaptts_amd.src.model_trainers.AcousticModelTrainer import AcousticModelTrainer
    logging.getLogger().setLevel(logging.WARNING)
    ids, wdir, qdir, g = materialise(os.path.join(ROOT, "tests", "golden"), os.path.join(root, "data%d" % rank))
    hp = AcousticModelTrainer.create_hparams()
    hp.num_questions, hp.voice, hp.frame_size_ms, hp.num_coded_sps = 409, "full", 5, 20
    hp.out_dir = os.path.join(root, "out_%d" % int(cache))        # (shared by the ranks: rank 0 writes, all read)
    hp.seed, hp.epochs, hp.use_gpu, hp.num_gpus = 1, 3, True, world
    hp.dataset_num_workers_gpu = 2
    hp.model_type = "RNNDYN-1_RELU_32-1_FC_67"
    hp.batch_size_train, hp.batch_size_val = 4, 50
    hp.optimiser_args["lr"] = 0.001
    hp.model_name, hp.world_dir = "dp_model", wdir
    hp.epochs_per_checkpoint = 1000
    hp.val_set_perc, hp.test_set_perc = 0.25, 0.0        # (two validation utterances: a batch holds >= one per rank, :392-395)
    hp.dataset_device_cache = cache
    trainer = AcousticModelTrainer(**AcousticModelTrainer.legacy_support_init(wdir, qdir, ids, hp.num_questions, hp))
    trainer.init(hp)
    val, train, handler = trainer.train(hp)
    torch.cuda.synchronize()
    name = next(iter(train))
    loader = handler.dataloader_train
    ret[rank] = ([float(v) for v in train[name]], [float(v) for v in val[name]],
                 torch.cat([p.detach().reshape(-1) for p in handler.model.parameters()]).cpu().numpy(),
                 dict(getattr(loader, "stats", {})), type(loader).__name__)
    dist.barrier()
    dist.destroy_process_group()


def test_trainer_with_the_device_batch_cache_under_two_ranks(gpu, tmp_path):
    """AcousticModelTrainer.train under data parallelism (reference: ModularModelHandlerPyTorch.py:732-735, the loop
    :683-882): with hparams.dataset_device_cache every rank caches the utterances of ITS shard of each global batch as
    the shuffling brings them round -- the same per-epoch losses and the same parameters on both ranks as the run that
    reads every utterance again each epoch (the sampler's seed is rank 0's in both)."""
    import torch.multiprocessing as mp
    runs = {}
    for cache in (False, True):
        ret = mp.get_context("spawn").Manager().dict()
        torch.manual_seed(123)
        mp.spawn(_trainer_dp_worker, args=(2, _free_port(), ret, str(tmp_path), cache), nprocs=2, join=True)
        runs[cache] = {r: ret[r] for r in (0, 1)}
    for cache in (False, True):
        assert np.array_equal(runs[cache][0][2], runs[cache][1][2])            # ranks in lock step
        assert runs[cache][0][0] == runs[cache][1][0]                          # global losses on both
    assert runs[True][0][4] == "CachedBatchLoader" and runs[False][0][4] != "CachedBatchLoader"
    st = runs[True][0][3]
    assert st["misses"] > 0 and st["hits"] > 0 and st["passed_through"] == 0
    # rank 0 draws the sampler's seed from the generator the trainer has just seeded (hparams.seed): both runs walk the
    # same batches in the same order, and the cached batches are prepare_batch's bit for bit -- so are the trajectories
    assert runs[True][0][0] == runs[False][0][0] and runs[True][0][1] == runs[False][0][1]
    assert float(np.abs(runs[True][0][2] - runs[False][0][2]).max()) <= 1e-7
    assert runs[True][0][0][-1] < runs[True][0][0][0]
