"""Data-parallel plumbing (one process per GPU, torch.distributed: 'nccl' = RCCL over xGMI on
the GPU box, 'gloo' in CPU tests).  The reference's multi-GPU hook is torch.nn.DataParallel
(ModularModelHandlerPyTorch.py:732-735), single process and non-functional;
the semantics defined here: synchronous DP whose result equals the single-GPU step on the
concatenated batch.

* utterances are sharded, never split (feature extraction / MLPG / synthesis need no collective);
* the loss is sum(masked sq. err) / (GLOBAL valid frames * D) (NamedLoss 'mean_per_frame',
  loss/NamedLoss.py:113-117), so local gradients are scaled by the global frame count BEFORE a
  sum all-reduce of the flat gradient buffer -- no second pass, no averaging;
* normalisation statistics are additive (MeanStdDevExtractor.combine_stats :163-204): one
  all-reduce(sum) of (count, sum x, sum x^2 | sum x x^T).
"""
import os

import numpy as np
import torch
import torch.distributed as dist

# A world of ONE rank normally skips every collective.  With this switch on (ITTS_FORCE_DIST=1 or
# force_collectives(True)) they are issued anyway -- a one-rank sum / broadcast is the identity --
# which is how the RCCL path (library load, communicator, stream ordering against the kernels'
# streams, teardown) is exercised on a one-GPU box: tests/test_gpu_dp.py, `bench.py --force-dist`.
_FORCE = os.environ.get("ITTS_FORCE_DIST", "0") == "1"


def force_collectives(on=True):
    global _FORCE
    _FORCE = bool(on)


def _active(group=None):
    """True when a collective has to be issued: a process group exists and it has more than one
    rank (or the one-rank switch is on)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return _FORCE or dist.get_world_size(group) > 1


def shard_by_length(lengths, world_size):
    """Greedy length-balanced partition of utterance indices (longest first onto the lightest
    rank). Deterministic; returns world_size lists of indices (each sorted ascending)."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    loads = [0] * world_size
    shards = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (loads[k], k))
        shards[r].append(i)
        loads[r] += int(lengths[i])
    return [sorted(s) for s in shards]


def global_sum(value, group=None, device=None):
    """Sum of a python number over all ranks (e.g. the global valid-frame count of a step)."""
    if not _active(group):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return float(t.item())


def allreduce_flat_(buf, group=None, async_op=False):
    """In-place sum all-reduce of a flat buffer (gradients or statistics).  async_op: returns the
    work handle (None when no collective is needed); the caller waits before it reads `buf`."""
    if _active(group):
        work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        if async_op:
            return work
    return None if async_op else buf


class NativeComm:
    """RCCL communicator held by libidiaptts_amd itself (include/idiaptts_amd.h: itts_comm_* /
    itts_allreduce_flat) -- what a maintainer binding only the C ABI uses for the gradient exchange
    that torch.nn.DataParallel does in the reference (ModularModelHandlerPyTorch.py:732-735).  The
    128-byte id travels from rank 0 over `broadcast_bytes` (default: torch.distributed's store of
    the existing process group; a one-rank communicator needs none).  The device is the current one."""

    def __init__(self, rank=0, world=1, broadcast_bytes=None):
        import ctypes
        from . import lib
        self._lib = lib
        self.comm = None
        L = lib.load()
        ident = ctypes.create_string_buffer(128)
        if rank == 0:
            lib.check(L.itts_comm_unique_id(ctypes.cast(ident, ctypes.c_void_p)), "itts_comm_unique_id")
        if world > 1:
            if broadcast_bytes is None:
                def broadcast_bytes(b):
                    obj = [bytes(b) if rank == 0 else None]
                    dist.broadcast_object_list(obj, src=0)
                    return obj[0]
            ident = ctypes.create_string_buffer(broadcast_bytes(ident.raw), 128)
        comm = ctypes.c_void_p()
        lib.check(L.itts_comm_init_rank(ctypes.cast(ident, ctypes.c_void_p), int(world), int(rank),
                                        ctypes.byref(comm)), "itts_comm_init_rank")
        self.comm, self.rank, self.world = comm, rank, world

    def allreduce_flat_(self, buf, op="sum", stream=None):
        """In-place reduction of a contiguous float32 / float64 device tensor over the ranks,
        asynchronous on `stream` (default: torch's current stream)."""
        lib = self._lib
        assert buf.is_cuda and buf.is_contiguous() and buf.dtype in (torch.float32, torch.float64)
        s = stream.cuda_stream if stream is not None else lib.current_stream()
        lib.check(lib.load().itts_allreduce_flat(
            buf.data_ptr(), buf.numel(), 0 if buf.dtype == torch.float32 else 1,
            {"sum": 0, "max": 1, "avg": 2}[op], self.comm, s), "itts_allreduce_flat")
        return buf

    def close(self):
        if self.comm is not None:
            comm, self.comm = self.comm, None
            self._lib.check(self._lib.load().itts_comm_destroy(comm), "itts_comm_destroy")

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):            # a communicator dropped without close() (an exception on the way) is not leaked
        try:
            self.close()
        except Exception:
            pass


def allreduce_stats_(extractor, group=None, device=None):
    """Merges Mean{StdDev,Covariance}Extractor statistics across ranks in place.  A rank whose
    shard was empty (no add_sample yet: scalar zeros) takes the shapes of the others."""
    second = "sum_squared_frames" if hasattr(extractor, "sum_squared_frames") \
        else "sum_product_frames"
    a = np.atleast_1d(np.asarray(extractor.sum_frames, dtype=np.float64))
    b = np.atleast_1d(np.asarray(getattr(extractor, second), dtype=np.float64))
    if _active(group):
        dims = torch.tensor([a.shape[-1], b.ndim], dtype=torch.int64)
        if device is not None:
            dims = dims.to(device)
        dist.all_reduce(dims, op=dist.ReduceOp.MAX, group=group)
        d, nd = (int(v) for v in dims.cpu())
        if extractor.sum_length == 0 and a.shape[-1] != d:
            a = np.zeros((1, d) if nd == 2 else (d,))     # the covariance extractor keeps [1, D] sums
            b = np.zeros((d, d) if nd == 2 else (d,))
    flat = torch.from_numpy(np.concatenate([[float(extractor.sum_length)], a.ravel(), b.ravel()]))
    if device is not None:
        flat = flat.to(device)
    allreduce_flat_(flat, group)
    flat = flat.cpu().numpy()
    extractor.sum_length = int(round(flat[0]))
    extractor.sum_frames = flat[1:1 + a.size].reshape(a.shape)
    setattr(extractor, second, flat[1 + a.size:].reshape(b.shape))
    return extractor


def dp_rank_world(group=None):
    """(rank, world size) of the data-parallel group, (0, 1) without torch.distributed."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def allreduce_module_grads_(params, local_weight, group=None):
    """Gradient step of synchronous data parallelism for an autograd module whose loss is a mean
    over the LOCAL valid frames: every rank scales its gradients by local_weight = n_local /
    n_global, then one sum all-reduce of the flattened gradients (a single bucket: the acoustic
    models here hold a few million parameters).  The result equals the gradient of the
    single-process step on the concatenated batch."""
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    if not _active(group):
        return
    flat = torch._utils._flatten_dense_tensors(grads)
    flat.mul_(float(local_weight))
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    for g, synced in zip(grads, torch._utils._unflatten_dense_tensors(flat, grads)):
        g.copy_(synced)


def broadcast_int(value, src=0, group=None, device=None):
    """The integer `value` of rank `src` on every rank (e.g. the seed of the epoch's shuffling)."""
    if not _active(group):
        return int(value)
    t = torch.tensor([int(value)], dtype=torch.int64, device=device)
    dist.broadcast(t, src=src, group=group)
    return int(t.item())


def broadcast_tensors_(tensors, src=0, group=None):
    """In-place broadcast of a list of tensors from rank `src` (model parameters, buffers,
    optimiser state after create / load): one flat message per dtype and device."""
    if not _active(group):
        return
    # RCCL moves device memory only, gloo in these tests host memory only: tensors that live on the
    # other side (torch.optim.Adam keeps state['step'] on the host) travel through a staging copy
    nccl = dist.get_backend(group) == "nccl"
    stage_dev = torch.device("cuda", torch.cuda.current_device()) if nccl else torch.device("cpu")
    buckets = {}
    for t in tensors:
        if torch.is_tensor(t) and t.numel() > 0:
            buckets.setdefault((t.dtype, t.device), []).append(t)
    for (dtype, device), ts in buckets.items():
        flat = torch._utils._flatten_dense_tensors([t.detach() for t in ts])
        moved = (device.type == "cuda") != nccl
        wire = flat.to(stage_dev) if moved else flat
        dist.broadcast(wire, src=src, group=group)
        if moved:
            flat = wire.to(device)
        for t, synced in zip(ts, torch._utils._unflatten_dense_tensors(flat, ts)):
            t.detach().copy_(synced)


def barrier(group=None):
    if _active(group):
        dist.barrier(group=group)
