"""Model handler: batching, the train / validation loop, inference, checkpoints -- the reference's
ModularModelHandlerPyTorch (neural_networks/pytorch/ModularModelHandlerPyTorch.py) with the same
method names and argument meaning:

  prepare_batch / sequence_mask / unsorted_pad_sequence   (:388-499)
  set_dataset / _get_dataloader                           (:500-548)
  set_optimiser (Adam, SGD) / set_scheduler / run_scheduler  (:553-656, :941-962)
  train / test / process_dataloader                       (:667-882)
  inference                                               (:964-993)
  save_checkpoint / load_checkpoint                       (:71-262; `params_<suffix>`,
      `optimiser_<suffix>`, `scheduler_<suffix>`, `config.json`, suffix in {e<N>, s<N>, best,
      last}; each file `torch.save({'params': state_dict, 'epoch', 'step'[, 'best_loss']})`)

The optimisers are fused HIP kernels (one launch per tensor), EMA is a HIP kernel, and the
models / losses the handler drives run the HIP forward / backward; `process_batch` is one
iteration of process_dataloader for callers that bring their own batches."""
import glob
import logging
import os
import re
from datetime import datetime
from functools import partial

import numpy as np
import torch
from torch.nn.utils.rnn import pad_sequence
from torch.optim.lr_scheduler import ExponentialLR, LambdaLR, ReduceLROnPlateau
from torch.utils.data import DataLoader

from idiaptts_amd import ops, parallel
from idiaptts_amd.misc import logging_sinks
from idiaptts_amd.nn.functional import padding_rows_identical, unit_gradient
from idiaptts_amd.src.neural_networks.pytorch import config_json


class HipAdam(torch.optim.Optimizer):
    """torch.optim.Adam semantics (no amsgrad) on the HIP kernels.

    Flat arena (default on the GPU): the parameters of a group, their gradients and both moment
    buffers live in four flat fp32 buffers; `p.data`, `p.grad` and the per-parameter state tensors
    are views into them (the state_dict stays torch.optim.Adam's).  One step is then ONE launch
    per group that also does what the reference's handler does around the optimiser
    (ModularModelHandlerPyTorch.py:810-831): gradient clipping by norm (norm types 2 and inf) and
    by value in front, the parameter EMA (ExponentialMovingAverage.py:32-45) behind -- see
    `configure_clipping` / `attach_ema`; the data-parallel all-reduce runs on the flat gradient
    without a flatten / unflatten copy (`allreduce_grads_`).  Autograd accumulates into the
    gradient views in place, so `zero_grad` zeroes the flat buffer instead of dropping `.grad`.
    Differences to torch: a parameter that never receives a gradient is stepped with g = 0
    (torch skips it)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                 flat=True):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._arenas = None
        self._clip = None            # (norm_kind | None, max_norm, clip_value | None)
        self._ema = None
        self._norm_accum = None
        self._overlap = None         # state of the bucket reductions started during backward
        self.last_overlap = None
        self._flat = flat
        self._build_arenas()

    # ------------------------------------------------------------------------------ flat arena
    def _build_arenas(self):
        """(Re)creates the flat buffers; a no-op while the parameters are not (yet) fp32 on a GPU
        -- tried again at the next zero_grad / step, e.g. after `model.cuda()`."""
        if not self._flat:
            return
        groups = [[p for p in g['params'] if p.requires_grad] for g in self.param_groups]
        ok = all(p.is_cuda and p.dtype == torch.float32 and not p.is_sparse
                 for ps in groups for p in ps) and any(len(ps) for ps in groups)
        if not ok:
            return
        self._arenas = []
        for ps in groups:
            if not ps:
                self._arenas.append(None)
                continue
            n = sum(p.numel() for p in ps)
            dev = ps[0].device
            a = dict(params=ps, n=n, step=0,
                     p=torch.empty(n, dtype=torch.float32, device=dev),
                     g=torch.zeros(n, dtype=torch.float32, device=dev),
                     m=torch.zeros(n, dtype=torch.float32, device=dev),
                     v=torch.zeros(n, dtype=torch.float32, device=dev), shadow=None)
            off = 0
            with torch.no_grad():
                for p in ps:
                    k = p.numel()
                    a['p'][off:off + k].copy_(p.data.reshape(-1))
                    p.data = a['p'][off:off + k].view(p.shape)
                    if p.grad is not None:
                        a['g'][off:off + k].copy_(p.grad.reshape(-1))
                    p.grad = a['g'][off:off + k].view(p.shape)
                    st = self.state[p]
                    if 'exp_avg' in st:      # re-pack (load_state_dict)
                        a['m'][off:off + k].copy_(st['exp_avg'].reshape(-1))
                        a['v'][off:off + k].copy_(st['exp_avg_sq'].reshape(-1))
                        a['step'] = max(a['step'], int(st.get('step', 0)))
                    st['step'] = a['step']
                    st['exp_avg'] = a['m'][off:off + k].view(p.shape)
                    st['exp_avg_sq'] = a['v'][off:off + k].view(p.shape)
                    off += k
            self._arenas.append(a)
        if self._ema is not None:
            self.attach_ema(self._ema)

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)       # replaces the state tensors by copies
        if self._arenas is not None:
            self._build_arenas()

    def zero_grad(self, set_to_none=True):
        if self._arenas is None:
            self._build_arenas()
        if self._arenas is None:
            return super().zero_grad(set_to_none)
        for a in self._arenas:
            if a is not None:
                a['g'].zero_()

    def configure_clipping(self, norm_type=None, max_norm=None, clip_value=None):
        """Gradient clipping inside the step (torch.nn.utils.clip_grad_norm_ then
        clip_grad_value_).  Returns False when this optimiser cannot do it (no arena, or a norm
        type other than 2 / inf): the caller then clips with the torch utilities before step()."""
        self._clip = None
        if norm_type is None and clip_value is None:
            return True
        kind = None
        if norm_type is not None:
            if float(norm_type) == 2.0:
                kind = 2
            elif float(norm_type) == float("inf"):
                kind = 0
            else:
                return False
        if self._arenas is None:
            return False
        self._clip = (kind, float(max_norm) if kind is not None else 0.0, clip_value)
        return True

    def attach_ema(self, ema):
        """Fuses the update of an ExponentialMovingAverage into the step: its shadow parameters
        are re-pointed into a flat buffer in this optimiser's parameter order.  False without an
        arena or when the two models do not line up."""
        if ema is None:
            return False
        self._ema = ema                 # remembered: attached when the arena comes into being
        if self._arenas is None:
            self._build_arenas()
            return self._arenas is not None and ema.fused
        by_param = {id(p): n for n, p in ema.source_named_parameters()}
        for a in self._arenas:
            if a is None:
                continue
            names = [by_param.get(id(p)) for p in a['params']]
            if any(n is None or n not in ema.shadow for n in names):
                return False
            a['shadow'] = torch.empty_like(a['p'])
            off = 0
            for n, p in zip(names, a['params']):
                k = p.numel()
                a['shadow'][off:off + k].copy_(ema.shadow[n].reshape(-1))
                ema.repoint(n, a['shadow'][off:off + k].view(p.shape))
                off += k
        ema.fused = True
        return True

    def allreduce_grads_(self, local_weight, group=None):
        """Data-parallel gradient step on the flat buffers (see parallel.allreduce_module_grads_);
        False without an arena.  After `begin_overlapped_allreduce` this only waits for the bucket
        reductions that backward has already started and reduces what is left."""
        if self._arenas is None:
            return False
        ov, self._overlap = self._overlap, None
        for ai, a in enumerate(self._arenas):
            if a is None:
                continue
            if ov is None:
                a['g'].mul_(float(local_weight))
                parallel.allreduce_flat_(a['g'], group)
                continue
            for bi, (lo, hi, _) in enumerate(a['buckets']):
                if (ai, bi) not in ov['issued']:      # a bucket some parameter of which got no gradient this step
                    a['g'][lo:hi].mul_(ov['weight'])
                    ov['work'].append(parallel.allreduce_flat_(a['g'][lo:hi], ov['group'], async_op=True))
        if ov is not None:
            for w in ov['work']:
                if w is not None:
                    w.wait()
            self.last_overlap = {"buckets_during_backward": len(ov['issued']), "collectives": len(ov['work'])}
        return True

    # -------------------------------------------------------------- communication beside backward
    # "overlap layer-3 gradient communication with layer-1/2 backward".  The flat gradient
    # arena is cut into buckets of whole parameters in arena order (>= `bucket_elems` values each: about a layer of the
    # recurrent models); a post-accumulate hook per parameter counts a bucket's gradients as autograd delivers them --
    # output layer first -- and the moment a bucket is complete its slice is scaled by this rank's share of the global
    # frame count and goes into an asynchronous sum all-reduce, while backward carries on with the layers in front.
    # Every rank builds the same graph, so the buckets complete -- and the collectives are issued -- in the same order.
    overlap_bucket_elems = 1 << 20        # values per bucket (about a layer of the recurrent models); tests lower it

    def begin_overlapped_allreduce(self, local_weight, group=None, bucket_elems=None):
        """Call between zero_grad() and backward(); `allreduce_grads_` afterwards waits.  False (and nothing armed)
        without an arena or outside a data-parallel run."""
        self._overlap = None
        if self._arenas is None or not parallel._active(group):
            return False
        bucket_elems = int(bucket_elems or self.overlap_bucket_elems)
        for ai, a in enumerate(self._arenas):
            if a is None or 'buckets' in a:
                continue
            buckets, lo, off, members = [], 0, 0, []
            for p in a['params']:
                members.append(p)
                off += p.numel()
                if off - lo >= bucket_elems:
                    buckets.append((lo, off, members))
                    lo, members = off, []
            if members:
                buckets.append((lo, off, members))
            a['buckets'] = buckets
            for bi, (_, _, ps) in enumerate(buckets):
                for p in ps:
                    p.register_post_accumulate_grad_hook(
                        lambda _p, ai=ai, bi=bi: self._bucket_hook(ai, bi))
        self._overlap = {"weight": float(local_weight), "group": group, "issued": set(), "work": [],
                         "count": {}}
        return True

    def _bucket_hook(self, ai, bi):
        ov = self._overlap
        if ov is None:
            return
        a = self._arenas[ai]
        lo, hi, ps = a['buckets'][bi]
        n = ov['count'].get((ai, bi), 0) + 1
        ov['count'][(ai, bi)] = n
        if n == len(ps) and (ai, bi) not in ov['issued']:
            ov['issued'].add((ai, bi))
            a['g'][lo:hi].mul_(ov['weight'])
            ov['work'].append(parallel.allreduce_flat_(a['g'][lo:hi], ov['group'], async_op=True))

    @torch.no_grad()
    def step(self, closure=None):
        if self._arenas is None:
            return self._step_per_tensor()
        accum, kind, max_norm, clip_value = None, 2, 0.0, 0.0
        if self._clip is not None:
            kind, max_norm, clip_value = self._clip
            if kind is not None:
                live = [a for a in self._arenas if a is not None]
                if self._norm_accum is None:
                    self._norm_accum = torch.zeros(1, dtype=torch.float32, device=live[0]['p'].device)
                accum = self._norm_accum
                for i, a in enumerate(live):
                    ops.grad_norm_accum(a['g'], accum, kind, accumulate=i > 0)
        for group, a in zip(self.param_groups, self._arenas):
            if a is None:
                continue
            a['step'] += 1
            ops.adam_step_fused(a['p'], a['g'], a['m'], a['v'], a['step'], lr=group['lr'],
                                betas=group['betas'], eps=group['eps'],
                                weight_decay=group['weight_decay'], norm_accum=accum,
                                norm_kind=kind if kind is not None else 2,
                                clip_max_norm=max_norm, clip_value=clip_value or 0.0,
                                ema_shadow=a['shadow'],
                                ema_decay=self._ema.decay if a['shadow'] is not None else 0.0)
            for p in a['params']:
                self.state[p]['step'] = a['step']

    def _step_per_tensor(self):
        for group in self.param_groups:
            for p in group['params']:
                if p.grad is None:
                    continue
                state = self.state[p]
                if len(state) == 0:
                    state['step'] = 0
                    state['exp_avg'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    state['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                state['step'] = int(state['step']) + 1
                ops.adam_step(p.data.view(-1), p.grad.contiguous().view(-1),
                              state['exp_avg'].view(-1), state['exp_avg_sq'].view(-1),
                              state['step'], lr=group['lr'], betas=group['betas'],
                              eps=group['eps'], weight_decay=group['weight_decay'])


class HipSGD(torch.optim.Optimizer):
    """torch.optim.SGD semantics through itts_sgd_step."""

    def __init__(self, params, lr=1e-3, momentum=0.0, dampening=0.0, weight_decay=0.0,
                 nesterov=False):
        if nesterov and (momentum <= 0 or dampening != 0):
            raise ValueError("Nesterov momentum requires a momentum and zero dampening")
        super().__init__(params, dict(lr=lr, momentum=momentum, dampening=dampening,
                                      weight_decay=weight_decay, nesterov=nesterov))

    @torch.no_grad()
    def step(self, closure=None):
        for group in self.param_groups:
            for p in group['params']:
                if p.grad is None:
                    continue
                state = self.state[p]
                buf, first = None, False
                if group['momentum'] != 0:
                    first = 'momentum_buffer' not in state
                    if first:
                        state['momentum_buffer'] = torch.empty_like(
                            p, memory_format=torch.contiguous_format)
                    buf = state['momentum_buffer'].view(-1)
                ops.sgd_step(p.data.view(-1), p.grad.contiguous().view(-1), buf, first,
                             lr=group['lr'], momentum=group['momentum'],
                             dampening=group['dampening'], weight_decay=group['weight_decay'],
                             nesterov=group['nesterov'])


class ExponentialMovingAverage(object):
    """Shadow copy of the trainable parameters, `shadow = decay * shadow + (1 - decay) * x` after
    every optimiser step (reference ExponentialMovingAverage.py:13-45); validation and the saved
    checkpoints use the averaged parameters.  HipAdam.attach_ema moves the shadow parameters into
    a flat buffer and updates them inside its fused step (`fused` is then True and
    update_params a no-op)."""

    def __init__(self, model, decay):
        import copy
        self.model = copy.deepcopy(model)
        self.decay = decay
        self.shadow = {}
        self.fused = False
        self._source = model
        for name, param in self.model.named_parameters():
            if param.requires_grad:
                self.shadow[name] = param.data
            param.detach_()

    def source_named_parameters(self):
        return self._source.named_parameters()

    def repoint(self, name, tensor):
        """Makes `tensor` (same shape, already holding the values) the storage of shadow `name`."""
        dict(self.model.named_parameters())[name].data = tensor
        self.shadow[name] = tensor

    def update_params(self, other_model):
        assert other_model is not self.model
        if self.fused:
            return
        for name, param in other_model.named_parameters():
            if name in self.shadow:
                ops.ema_update(self.shadow[name].view(-1), param.data.contiguous().view(-1),
                               self.decay)


def _host_copy(obj):
    """Detached, compact CPU copies of all tensors in a (nested) state dict."""
    if torch.is_tensor(obj):
        return obj.detach().to("cpu", copy=True).contiguous().clone()
    if isinstance(obj, dict):
        return type(obj)((k, _host_copy(v)) for k, v in obj.items())
    if isinstance(obj, (list, tuple)):
        return type(obj)(_host_copy(v) for v in obj)
    return obj


class _CheckpointWriter(object):
    """One background thread that serialises checkpoints (hparams.async_checkpoint): the training
    loop only pays for the device -> host copy of the state, `torch.save` and the file system run
    behind the next steps.  Files appear in submission order;
    `wait()` (also called before any checkpoint is read and at interpreter exit) blocks until
    everything is on disk and re-raises a writer error."""

    def __init__(self):
        import atexit
        import queue
        import threading
        self._queue = queue.Queue()
        self._error = None
        self._thread = threading.Thread(target=self._run, name="itts-checkpoint-writer",
                                        daemon=True)
        self._thread.start()
        atexit.register(self.wait)

    def _run(self):
        while True:
            obj, path = self._queue.get()
            try:
                torch.save(obj, path + ".tmp")
                os.replace(path + ".tmp", path)
            except Exception as e:       # surfaced by wait()
                self._error = e
            finally:
                self._queue.task_done()

    def submit(self, obj, path):
        self._queue.put((_host_copy(obj), path))

    def wait(self):
        self._queue.join()
        if self._error is not None:
            e, self._error = self._error, None
            raise e


class ModularModelHandlerPyTorch(object):
    logger = logging.getLogger(__name__)
    _checkpoint_writer = None          # shared _CheckpointWriter, created on first use

    @classmethod
    def wait_for_checkpoints(cls):
        """Blocks until all asynchronously written checkpoints are on disk."""
        if cls._checkpoint_writer is not None:
            cls._checkpoint_writer.wait()

    def __init__(self):
        self.model = None
        self.model_config = None
        self.optimiser = None
        self.scheduler = None
        self._scheduler_step_fn = None
        self.ema = None
        self.losses = []
        self.dataloader_train = None
        self.dataloader_val = None
        self._resident = None          # HBM-resident training state (set_dataset, resident_dataset)
        self._cached_loaders = {}      # loaders that keep their utterances' rows in HBM (DeviceBatchCache)
        self._stock_padding = False    # the loaders' batches come from prepare_batch: padding positions are zeros
        self._batch_guard = None       # deferred NaN guard of process_batch(blocking=False)

    @staticmethod
    def cuda_is_available():
        return torch.cuda.is_available()

    @staticmethod
    def device_count():
        return torch.cuda.device_count()

    @staticmethod
    def seed(seed):
        torch.manual_seed(seed)

    # ----------------------------------------------------------------------------- batching
    @staticmethod
    def sequence_mask(sequence_length, max_len=None, batch_first=False):
        """[B, T, 1] / [T, B, 1] float mask, 1 where t < length (reference :467-491)."""
        sequence_length = torch.as_tensor(sequence_length)
        if max_len is None:
            max_len = int(sequence_length.max())
        t = torch.arange(0, int(max_len), dtype=torch.long)
        length = sequence_length.to(torch.long)
        m = (t.unsqueeze(0) < length.unsqueeze(1)) if batch_first else \
            (t.unsqueeze(1) < length.unsqueeze(0))
        return m.unsqueeze(-1).contiguous().float()

    @staticmethod
    def unsorted_pad_sequence(sequence, batch_first, pinned=False):
        """pad_sequence; pinned: straight into page-locked memory (the padded batch is 90 MB at 32 utterances:
        padding into pageable memory and page-locking a copy afterwards is one pass over it more)."""
        sequence = [torch.from_numpy(s) if isinstance(s, np.ndarray) else s for s in sequence]
        if not pinned or not torch.cuda.is_available() or any(s.is_cuda or s.dim() == 0 for s in sequence):
            return pad_sequence(sequence, batch_first)
        max_len = max(s.shape[0] for s in sequence)
        trailing = tuple(sequence[0].shape[1:])
        shape = ((len(sequence), max_len) if batch_first else (max_len, len(sequence))) + trailing
        out = torch.zeros(shape, dtype=sequence[0].dtype, pin_memory=True)
        for i, s in enumerate(sequence):
            if batch_first:
                out[i, :s.shape[0]] = s
            else:
                out[:s.shape[0], i] = s
        return out

    @staticmethod
    def prepare_batch(batch, common_divisor=1, batch_first=False, mask_keys=(), shard=None, pin_output=False):
        """Collate function (reference :388-465).  `batch` holds the dataset's
        ({name: array [T_i, D]}, dataset) items (bare dicts are accepted too).  The remainder that
        is not divisible by `common_divisor` (# GPUs) is dropped first (:392-395); every key whose
        data reader sets `requires_seq_mask` (or that is listed in `mask_keys`) also gets
        `<key>_mask`; keys without a reader or with ragged non-array content stay lists.
        `shard=(rank, world)` keeps this rank's samples of the (identically drawn) global batch.
        Returns (data, temporal lengths)."""
        assert len(batch) >= common_divisor
        remainder = len(batch) % common_divisor
        if remainder > 0:
            batch = batch[:-remainder]
        if shard is not None:            # data parallel: rank r of w keeps samples r, r + w, ...
            batch = batch[shard[0]::shard[1]]
        dataset = None
        if isinstance(batch[0], (tuple, list)):
            dataset = batch[0][1]
            batch = [b[0] for b in batch]
        data, lengths = dict(), dict()
        for key in batch[0].keys():
            values = [b[key] for b in batch if key in b]
            if key == "_id_list":
                data[key] = list(values)
                continue
            reader = None
            if dataset is not None:
                try:
                    reader = dataset.get_datareader_by_output_name(key)
                except KeyError:
                    data[key] = list(values)
                    continue
            shapes = torch.tensor([x.shape for x in values], dtype=torch.long)
            lengths[key] = shapes[:, 0]
            max_shape, max_idx = shapes.max(dim=0)
            max_frames = int(max_shape[0])
            if reader is not None and reader.min_frames is not None \
                    and max_frames < reader.min_frames:
                # Padding the longest sample brings the whole batch to min_frames.
                i = int(max_idx[0])
                padding = [(0, reader.min_frames - max_frames)] + \
                    [(0, 0)] * (values[i].ndim - 1)
                values[i] = reader.pad(values[i], padding)
                max_frames = reader.min_frames
            if reader is not None and reader.other_pad_dims is not None:
                for idx, sample in enumerate(values):
                    padding = [(0, 0)] * sample.ndim
                    for dim in reader.other_pad_dims:
                        if dim != 0:
                            padding[dim] = (0, int(max_shape[dim]) - sample.shape[dim])
                    values[idx] = reader.pad(sample, padding)
            if key in mask_keys or (reader is not None and reader.requires_seq_mask):
                assert reader is None or reader.other_pad_dims is None, \
                    "Sequence mask for padding in multiple dimensions is not implemented."
                data[key + "_mask"] = ModularModelHandlerPyTorch.sequence_mask(
                    lengths[key], max_frames, batch_first=batch_first)
                lengths[key + "_mask"] = lengths[key]
            data[key] = ModularModelHandlerPyTorch.unsorted_pad_sequence(values, batch_first, pinned=pin_output)
        return data, lengths

    def set_dataset(self, hparams, dataset_train, dataset_val, collate_fn=None):
        if hparams.get_value("resident_dataset", False):
            return self._set_resident_dataset(hparams, dataset_train, dataset_val, collate_fn)
        self._resident = None
        self._set_loaders(hparams, dataset_train, dataset_val, collate_fn)

    def _set_loaders(self, hparams, dataset_train, dataset_val, collate_fn=None):
        num_workers = hparams.dataset_num_workers_gpu if hparams.use_gpu \
            else hparams.dataset_num_workers_cpu
        # hparams.dataset_device_cache (default on, GPU only): what the readers return for an utterance stays in HBM
        # after its first use and later batches are gathered there (data_preparation/DeviceBatchCache.py)
        device_cache = bool(hparams.use_gpu and hparams.get_value("dataset_device_cache", True)
                            and torch.cuda.is_available())
        common = dict(batch_first=hparams.batch_first, collate_fn=collate_fn,
                      common_divisor=hparams.num_gpus, num_workers=num_workers,
                      pin_memory=hparams.dataset_pin_memory,
                      worker_kind=hparams.get_value("dataset_worker_kind", "thread"),
                      device_cache=device_cache,
                      device_cache_bytes=hparams.get_value("dataset_device_cache_bytes", None),
                      host_cache_bytes=hparams.get_value("dataset_host_cache_bytes", None))
        # loaders of datasets that are gone give their rows back
        live = {id(ds) for ds in (dataset_train, dataset_val) if ds is not None}
        for key in [k for k in self._cached_loaders if k[0] not in live]:
            del self._cached_loaders[key]
        self.dataloader_train = self._get_dataloader(
            batch_size=hparams.batch_size_train, dataset=dataset_train,
            shuffle=hparams.shuffle_train_set, **common)
        self.dataloader_val = self._get_dataloader(
            batch_size=hparams.batch_size_val, dataset=dataset_val,
            shuffle=hparams.shuffle_val_set, **common)
        # every padding position of a stock batch holds zeros (pad_sequence; a reader that pads up to `min_frames`
        # in another mode is the exception): the frame-independent layers may then skip them (nn/functional.py)
        stock = collate_fn is None or collate_fn is self.prepare_batch \
            or getattr(collate_fn, "__func__", None) is ModularModelHandlerPyTorch.prepare_batch
        self._stock_padding = bool(stock) and all(
            r.min_frames is None or getattr(r, "pad_mode", "constant") == "constant"
            for ds in (dataset_train, dataset_val) if ds is not None
            for r in getattr(ds, "datareaders", ()))

    @staticmethod
    def _items_draw_random_numbers(dataset):
        """True unless the dataset is known not to use a random generator in `__getitem__` (readers with `max_frames`
        may pick their window at random: data_preparation/PyTorchDatareadersDataset.py)."""
        readers = getattr(dataset, "datareaders", None)
        if readers is None:
            return True
        return any(getattr(r, "max_frames", None) is not None for r in readers)

    def _get_dataloader(self, batch_size, dataset, batch_first=True, collate_fn=None,
                        common_divisor=1, num_workers=1, pin_memory=True, shuffle=False,
                        worker_kind="process", device_cache=False, device_cache_bytes=None, host_cache_bytes=None):
        stock = collate_fn is None or collate_fn is self.prepare_batch
        collate_fn = self.prepare_batch if collate_fn is None else collate_fn
        rank, world = parallel.dp_rank_world()
        extra = {}
        generator = None
        cacheable = device_cache and stock and dataset is not None and len(dataset) > 0 \
            and hasattr(dataset, "get_datareader_by_output_name") and not self._items_draw_random_numbers(dataset)
        if cacheable:
            # the loader of an earlier set_dataset call over the same dataset keeps what it has uploaded
            # (ModularTrainer.train / test call set_dataset every time)
            key = (id(dataset), len(dataset), batch_size, bool(shuffle), bool(batch_first),
                   max(common_divisor, world), rank, world, num_workers if worker_kind == "thread" else 0)
            kept = self._cached_loaders.get(key)
            if kept is not None and kept.dataset is dataset:
                return kept
        if world > 1:
            # One process per GPU: every rank draws the same global batches and keeps its own
            # samples; common_divisor makes the split even.  "The same batches" must not depend on
            # the user seeding every rank alike: the sampler's generator is seeded with a value
            # rank 0 draws and broadcasts.
            common_divisor = max(common_divisor, world)
            extra["shard"] = (rank, world)
            seed = parallel.broadcast_int(int(torch.empty((), dtype=torch.int64).random_().item()),
                                          device=self._dist_device())
            generator = torch.Generator().manual_seed(seed)
        if cacheable:
            from idiaptts_amd.src.data_preparation.DeviceBatchCache import CachedBatchLoader
            device = self._device() if self.model is not None else torch.device("cuda", torch.cuda.current_device())
            loader = CachedBatchLoader(dataset, batch_size, shuffle, device, batch_first,
                                       common_divisor=common_divisor, shard=extra.get("shard"),
                                       threads=num_workers if worker_kind == "thread" else 0, generator=generator,
                                       byte_budget=device_cache_bytes, host_collate=self.prepare_batch,
                                       host_byte_budget=host_cache_bytes)
            self._cached_loaders[key] = loader
            return loader
        collate = partial(collate_fn, common_divisor=common_divisor, batch_first=batch_first, **extra)
        if num_workers > 0 and worker_kind == "thread" and not self._items_draw_random_numbers(dataset):
            # hparams.dataset_num_workers_*: readers in THREADS of this process (hparams.dataset_worker_kind =
            # "process" for torch's forked workers): the same batches in the same order, same draws from the global
            # generator; see ThreadedBatchLoader
            from idiaptts_amd.src.data_preparation.ThreadedBatchLoader import ThreadedBatchLoader
            import inspect
            if pin_memory and torch.cuda.is_available() and \
                    "pin_output" in inspect.signature(collate_fn).parameters:
                collate = partial(collate, pin_output=True)      # pad straight into page-locked memory
            return ThreadedBatchLoader(dataset, batch_size, shuffle, collate, threads=num_workers,
                                       generator=generator, pin_memory=pin_memory)
        return DataLoader(dataset=dataset, batch_size=batch_size, shuffle=shuffle,
                          num_workers=num_workers, generator=generator,
                          collate_fn=collate,
                          pin_memory=pin_memory and torch.cuda.is_available())

    # ------------------------------------------------------------------- HBM-resident data
    class _IndexDataset(torch.utils.data.Dataset):
        def __init__(self, n):
            self.n = n

        def __len__(self):
            return self.n

        def __getitem__(self, i):
            return i

    def _set_resident_dataset(self, hparams, dataset_train, dataset_val, collate_fn=None):
        """hparams.resident_dataset: the data readers run ONCE per id, their normalised, length
        matched outputs are packed into FrameShards and uploaded; the
        loaders then only draw utterance indices -- same batch sizes, same shuffling and the same
        RNG consumption as the per-item loaders -- and every mini-batch is gathered on the device
        as packed valid frames for the flat feed-forward step (idiaptts_amd.native_ff).  That step
        covers Linear groups without dropout, one input and one target stream, Adam, one unweighted
        masked MSE; any other model, optimiser or loss trains on the module path with the device
        batch cache instead (data_preparation/DeviceBatchCache.py), which holds the same rows."""
        from idiaptts_amd.native_ff import FlatFFModel
        from idiaptts_amd.src.data_preparation.FrameShard import FrameShard
        if not hparams.use_gpu:
            raise RuntimeError("hparams.resident_dataset needs hparams.use_gpu.")
        flat = FlatFFModel.from_module(self.model, self._device())
        reason = None
        if flat is None:
            reason = "the model is not a stack of Linear groups without dropout"
        else:
            try:
                for ds in (dataset_train, dataset_val):
                    if ds is not None:
                        self._resident_target_name(ds, self.model.input_names[0])
            except NotImplementedError as e:
                reason = str(e)
        if reason is not None:
            self.logger.info("resident_dataset: {}; training on the module path with the device batch cache."
                             .format(reason))
            self._resident = None
            return self._set_loaders(hparams, dataset_train, dataset_val, collate_fn)
        device = self._device()
        in_name = self.model.input_names[0]
        shards = {}
        for key, ds in (("train", dataset_train), ("val", dataset_val)):
            if ds is None:
                continue
            target = self._resident_target_name(ds, in_name)
            threads = hparams.dataset_num_workers_gpu \
                if hparams.get_value("dataset_worker_kind", "thread") == "thread" \
                and not self._items_draw_random_numbers(ds) else 0
            shards[key] = FrameShard.from_dataset(ds, in_name, target, threads=threads).to(device)
        common = dict(num_workers=0, collate_fn=list, pin_memory=False)
        self.dataloader_train = DataLoader(self._IndexDataset(len(shards["train"])),
                                           batch_size=hparams.batch_size_train,
                                           shuffle=hparams.shuffle_train_set, **common)
        self.dataloader_val = DataLoader(self._IndexDataset(len(shards["val"])),
                                         batch_size=hparams.batch_size_val,
                                         shuffle=hparams.shuffle_val_set, **common) \
            if "val" in shards else None
        self._resident = {"flat": flat, "shards": shards, "synced": False,
                          "datasets": (dataset_train, dataset_val, collate_fn)}

    def _resident_unsupported(self, hparams, training):
        """Why the flat feed-forward step cannot run this configuration (None: it can)."""
        if training and not isinstance(self.optimiser, HipAdam):
            return "the flat step trains with Adam"
        if hparams.grad_clip_norm_type is not None and \
                float(hparams.grad_clip_norm_type) not in (2.0, float("inf")):
            return "the flat step clips with norm type 2 or inf"
        if hparams.replace_inf_grads_by_zero:
            return "replace_inf_grads_by_zero is not part of the flat step"
        loss = self.losses[0] if len(self.losses) == 1 else None
        if loss is None or getattr(loss, "loss_weight", 1.0) != 1.0 or getattr(loss, "start_step", 0) > 0 \
                or getattr(loss, "kind", 0) != 0 or getattr(loss, "reduction", "mean_per_frame") != "mean_per_frame":
            return "the flat step trains one unweighted masked-MSE loss (mean_per_frame)"
        return None

    def _resident_fall_through(self, hparams, reason, dataloader):
        """Leaves the flat step for the module path (the rows come from the device batch cache from then on); returns
        the module path's loader that stands for `dataloader`."""
        self.logger.info("resident_dataset: {}; continuing on the module path with the device batch cache."
                         .format(reason))
        was_train = dataloader is self.dataloader_train
        self._resident_sync_to_module()
        train, val, collate_fn = self._resident["datasets"]
        self._resident = None
        self._set_loaders(hparams, train, val, collate_fn)
        return self.dataloader_train if was_train else self.dataloader_val

    @staticmethod
    def _resident_target_name(dataset, input_name):
        names = [n for r in dataset.datareaders for n in r.output_names if n != input_name]
        if len(names) != 1:
            raise NotImplementedError("resident_dataset expects one input and one target stream, "
                                      "found targets {}.".format(names))
        return names[0]

    def _resident_sync_from_module(self):
        """Flat buffers <- module parameters and optimiser state (fresh model or loaded
        checkpoint); done once before the first resident step."""
        res = self._resident
        flat = res["flat"]
        from idiaptts_amd.native_ff import FlatFFModel
        fresh = FlatFFModel.from_module(self.model, self._device())
        flat.params.copy_(fresh.params)
        flat.step_count = 0
        if self.optimiser is not None:
            from idiaptts_amd.nn.modules import LinearAct
            inner = getattr(self.model, "model", self.model)
            lin = [m for g in inner.layer_groups for m in g.module if isinstance(m, LinearAct)]
            for i, m in enumerate(lin):
                for p, view in ((m.weight, flat.weight), (m.bias, flat.bias)):
                    st = self.optimiser.state.get(p, {})
                    if "exp_avg" in st:
                        view(i, flat.exp_avg).copy_(st["exp_avg"])
                        view(i, flat.exp_avg_sq).copy_(st["exp_avg_sq"])
                        flat.step_count = int(st["step"])
        res["synced"] = True

    def _resident_sync_to_module(self):
        """Module parameters and optimiser state <- flat buffers (validation through the module
        stack, checkpoints, inference)."""
        res = self._resident
        if res is None or not res["synced"]:
            return
        flat = res["flat"]
        lin = flat.store_to_module(self.model)
        if res.get("ema_shadow") is not None and self.ema is not None:
            flat.store_to_module(self.ema.model, res["ema_shadow"])
        if self.optimiser is not None and isinstance(self.optimiser, HipAdam) and flat.step_count > 0:
            for i, m in enumerate(lin):
                for p, view in ((m.weight, flat.weight), (m.bias, flat.bias)):
                    self.optimiser.state[p] = {
                        "step": flat.step_count,
                        "exp_avg": view(i, flat.exp_avg).clone().contiguous(),
                        "exp_avg_sq": view(i, flat.exp_avg_sq).clone().contiguous()}
            if self.optimiser._arenas is not None:
                self.optimiser._build_arenas()        # re-pack the arena from the new state

    def _process_resident(self, dataloader, shard, hparams, total_epoch, total_steps,
                          current_epoch, training):
        """process_dataloader on HBM-resident shards with the flat feed-forward step
        (idiaptts_amd.native_ff): same losses, same Adam, same scheduler calls."""
        res = self._resident
        flat = res["flat"]
        clip_kind = None                    # (what the flat step cannot do never gets here: _resident_unsupported)
        if hparams.grad_clip_norm_type is not None:
            clip_kind = {2.0: 2, float("inf"): 0}[float(hparams.grad_clip_norm_type)]
        if not res["synced"]:
            self._resident_sync_from_module()
        if training and self.ema is not None and res.get("ema_shadow") is None:
            # flat shadow in the step's layout, seeded from the averaged model
            res["ema_shadow"] = torch.zeros_like(flat.params)
            from idiaptts_amd.native_ff import FlatFFModel
            res["ema_shadow"].copy_(FlatFFModel.from_module(self.ema.model, self._device()).params)
            self.ema.fused = True
        loss_name = self.losses[0].name
        rank, world = parallel.dp_rank_world()
        total = None
        scalars = logging_sinks.DeferredScalars(logging_sinks.open_scalar_writer(hparams))
        guard = logging_sinks.DeferredLossCheck(self._device(), check_inf=False)
        if hparams.log_memory_consumption:
            self.logger.info(logging_sinks.memory_message(hparams.use_gpu))
        for batch_index, indices in enumerate(dataloader):
            if world > 1:                      # this rank's utterances of the global batch
                indices = indices[:len(indices) - len(indices) % world][rank::world]
            x, y, lens = shard.gather(indices)
            n_local = float(lens.sum())
            n_global = parallel.global_sum(n_local, device=x.device)
            valid = torch.ones(x.shape[0], dtype=torch.uint8, device=x.device)
            if training:
                group = self.optimiser.param_groups[0]
                loss = flat.train_step(x, y, valid, n_global, lr=group["lr"], betas=group["betas"],
                                       eps=group["eps"], weight_decay=group["weight_decay"],
                                       world_size=world, clip_norm_kind=clip_kind,
                                       clip_max_norm=hparams.grad_clip_max_norm or 0.0,
                                       clip_value=hparams.grad_clip_thresh,
                                       ema_shadow=res.get("ema_shadow"),
                                       ema_decay=self.ema.decay if self.ema is not None else 0.0)
                total_steps += 1
            else:
                live = flat.params       # validation uses the averaged parameters (reference :704-707)
                if res.get("ema_shadow") is not None:
                    flat.params = res["ema_shadow"]
                try:
                    loss, _ = ops.masked_mse(flat.forward(x)[-1], y, valid, n_global,
                                             want_grad=False)
                finally:
                    flat.params = live
            loss = parallel.allreduce_flat_(loss.clone())[0]
            guard.submit({loss_name: loss})          # looked at one step late: no host stall per step
            if training:
                current_iter = self._get_current_iteration(
                    batch_index=batch_index, current_epoch=current_epoch,
                    dataloader_length=len(dataloader), hparams=hparams, total_epoch=total_epoch)
                self.run_scheduler(hparams=hparams, loss=loss, current_iter=current_iter)
                scalars.add_scalars("Train loss", {loss_name: loss}, total_steps)
            total = loss if total is None else total + loss
        guard.finish()
        mean = total / len(dataloader)
        if not training:
            scalars.add_scalars("Validation loss", {loss_name: mean}, total_steps)
        scalars.flush()
        return {loss_name: mean.cpu().numpy()}

    # -------------------------------------------------------------------------------- model
    def create_model(self, model_config, use_gpu=True):
        self.logger.info("Create network from config: {}".format(type(model_config)))
        self.model_config = model_config
        self.model = model_config.create_model()
        if use_gpu:
            self.model = self.model.cuda()
        self.sync_from_rank0()
        return self.model

    @staticmethod
    def _dist_device():
        """Device collectives of small host values run on: the current GPU under RCCL ('nccl'),
        host memory under gloo (None)."""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl":
            return torch.device("cuda", torch.cuda.current_device())
        return None

    def sync_from_rank0(self, optimiser=True):
        """Data parallelism (one process per GPU) needs identical replicas: after a model has been
        created or loaded every rank takes rank 0's parameters, buffers and optimiser state, so
        nothing depends on the ranks having been seeded alike.  No-op on a single process."""
        if parallel.dp_rank_world()[1] == 1 or self.model is None:
            return
        tensors = [p.data for p in self.model.parameters()] + [b for b in self.model.buffers()]
        if optimiser and self.optimiser is not None:
            # ONE object broadcast first: rank 0's optimiser state as a manifest -- per parameter (in
            # param_groups order, the same on every rank) the tensor entries with their shapes and the
            # python-valued entries (step counters of HipAdam / HipSGD), plus the arenas' counters.  The
            # other ranks shape their state after it (optimisers create state lazily: a rank that has
            # not stepped yet holds none), so that every rank then issues the SAME tensor broadcasts.
            import torch.distributed as dist
            rank = parallel.dp_rank_world()[0]
            opt = self.optimiser
            params = [p for g in opt.param_groups for p in g["params"]]
            arenas = getattr(opt, "_arenas", None) or []
            manifest = None
            if rank == 0:
                per_param = []
                for p in params:
                    entry = {}
                    for k, v in opt.state.get(p, {}).items():
                        entry[k] = ("tensor", tuple(v.shape), str(v.dtype).replace("torch.", ""), v.is_cuda) \
                            if torch.is_tensor(v) else ("value", v)
                    per_param.append(entry)
                manifest = (per_param, [None if a is None else a.get("step") for a in arenas])
            box = [manifest]
            dist.broadcast_object_list(box, src=0)
            per_param, arena_steps = box[0]
            for p, entry in zip(params, per_param):
                if not entry and p not in opt.state:
                    continue
                st = opt.state[p]
                for k in [k for k in st if k not in entry]:
                    del st[k]
                for k, spec in entry.items():
                    if spec[0] == "value":
                        st[k] = spec[1]
                    elif not (torch.is_tensor(st.get(k)) and tuple(st[k].shape) == spec[1]):
                        st[k] = torch.zeros(spec[1], dtype=getattr(torch, spec[2]),
                                            device=p.device if spec[3] else "cpu")
            for a, step in zip(arenas, arena_steps):
                if a is not None and step is not None:
                    a["step"] = step
            for p in params:
                tensors += [v for v in opt.state.get(p, {}).values() if torch.is_tensor(v)]
        # every tensor travels on the side the backend moves (RCCL: device memory, gloo: host
        # memory); broadcast_tensors_ stages the others (torch.optim keeps state['step'] on the host)
        parallel.broadcast_tensors_(tensors)
        if self._resident is not None:
            self._resident["synced"] = False

    def set_losses(self, losses):
        """Loss modules (the reference's signature, :550-551); loss configs are instantiated."""
        self.losses = [l.create_loss() if hasattr(l, "create_loss") else l for l in losses]

    def set_optimiser(self, hparams="Adam", reset=False, **optimiser_args):
        """reference :553-583.  Also accepts the optimiser type as a string with keyword
        arguments (`set_optimiser("Adam", lr=1e-3)`)."""
        if isinstance(hparams, str):
            cls = {"Adam": HipAdam, "SGD": HipSGD}.get(hparams)
            if cls is None:
                raise NotImplementedError("Optimiser type {} is not implemented.".format(hparams))
            self.optimiser = cls(self.model.parameters(), **optimiser_args)
            self._reattach_ema()
            return
        if self.optimiser is None or reset:
            if hparams.optimiser is None:
                self.logger.info("Create {} optimiser.".format(hparams.optimiser_type))
                if "params" in hparams.optimiser_args:
                    args = dict(hparams.optimiser_args)
                else:
                    if "lr" not in hparams.optimiser_args:       # backwards compatible
                        try:
                            hparams.optimiser_args["lr"] = hparams.learning_rate
                        except AttributeError:
                            raise AttributeError("Learning rate not defined in "
                                                 "hparams.optimiser_args[\"lr\"]")
                    args = {"params": self.model.parameters()}
                    args.update(hparams.optimiser_args)
                if hparams.optimiser_type == "Adam":
                    if args.pop("amsgrad", False):
                        raise NotImplementedError("amsgrad has no fused kernel.")
                    self.optimiser = HipAdam(**args)
                elif hparams.optimiser_type == "SGD":
                    self.optimiser = HipSGD(**args)
                else:
                    raise NotImplementedError("Optimiser type {} is not implemented."
                                              .format(hparams.optimiser_type))
            else:
                self.optimiser = hparams.optimiser(self.model.parameters())
            self._reattach_ema()
        if not hparams.use_saved_learning_rate and "lr" in hparams.optimiser_args:
            for g in self.optimiser.param_groups:
                g['lr'] = hparams.optimiser_args["lr"]

    def _reattach_ema(self):
        """A new optimiser means new flat buffers: an existing EMA either moves its shadow into
        them (fused update inside the step) or goes back to the separate update kernel -- it must
        never be left pointing at a discarded optimiser (its update would silently stop)."""
        if self.ema is None:
            return
        attach = getattr(self.optimiser, "attach_ema", None)
        if attach is None or not attach(self.ema):
            self.ema.fused = False

    def _reseed_ema(self):
        """After parameters were loaded the shadow restarts from them.  Deliberate deviation from the
        reference, which creates its EMA once from the model (ExponentialMovingAverage.py:13-30) and
        does not touch it in load_checkpoint: here checkpoints hold the AVERAGED parameters, so a
        shadow left at its pre-load values would pull the loaded model back towards the discarded
        one (tests/test_gpu_optim.py).  The flat copy of the shadow kept by the resident-dataset
        path is dropped with it and rebuilt from the reseeded EMA at the next step."""
        if self.ema is None:
            return
        with torch.no_grad():
            for name, p in self.model.named_parameters():
                if name in self.ema.shadow:
                    self.ema.shadow[name].copy_(p.data)
        if self._resident is not None:
            self._resident["ema_shadow"] = None

    def set_scheduler(self, hparams, current_epoch=None, current_step=None, reset=False):
        """reference :585-656: Plateau (stepped with the validation loss), Exponential and Noam
        (stepped per iteration unless epochs_per_scheduler_step is set)."""
        if self.scheduler is not None and not reset:
            return
        if hparams.scheduler is not None:
            self.scheduler = hparams.scheduler(self.optimiser)
            self._scheduler_step_fn = self._scheduler_step
            return
        if hparams.scheduler_type.lower() == "none":
            return
        assert hparams.scheduler_type != "default", \
            "Please define a default scheduler type in the trainer class."
        self.logger.info("Create {} scheduler.".format(hparams.scheduler_type))
        if current_epoch == 0:        # PyTorch schedulers count from -1 and step immediately.
            current_epoch = -1
        if current_step == 0:
            current_step = -1
        if hparams.scheduler_type == "Plateau":
            args = dict(hparams.scheduler_args)
            args.pop("verbose", None)
            self.scheduler = ReduceLROnPlateau(self.optimiser, **args)
            self._scheduler_step_fn = self._scheduler_step_with_loss
            if hparams.epochs_per_scheduler_step is None \
                    and hparams.iterations_per_scheduler_step is None:
                hparams.epochs_per_scheduler_step = 1
            return
        if current_step is None and self.dataloader_train is not None:
            current_step = max((current_epoch - 1) * len(self.dataloader_train), -1)
        per_iteration = hparams.epochs_per_scheduler_step is None
        if per_iteration and hparams.iterations_per_scheduler_step is None:
            hparams.iterations_per_scheduler_step = 1
        last = current_step if per_iteration else current_epoch - 1
        if last is not None and last >= 0:
            for group in self.optimiser.param_groups:
                group.setdefault('initial_lr', group['lr'])
        if hparams.scheduler_type == "Exponential":
            self.scheduler = ExponentialLR(self.optimiser, last_epoch=last,
                                           **hparams.scheduler_args)
        elif hparams.scheduler_type == "Noam":
            assert "wormup_steps" in hparams.scheduler_args, \
                "Please define wormup_steps in hparams.scheduler_args."
            warmup = float(hparams.scheduler_args['wormup_steps'])

            def noam_decay(iteration):
                return warmup ** 0.5 * np.minimum((iteration + 1) * warmup ** -1.5,
                                                  (iteration + 1) ** -0.5)
            self.scheduler = LambdaLR(self.optimiser, noam_decay, last_epoch=last)
        else:
            raise NotImplementedError("Scheduler type {} is not implemented."
                                      .format(hparams.scheduler_type))
        self._scheduler_step_fn = self._scheduler_step

    def _scheduler_step_with_loss(self, loss):
        self.scheduler.step(float(loss))

    def _scheduler_step(self, loss):
        self.scheduler.step()

    def run_scheduler(self, hparams, loss, current_iter=None, current_epoch=None):
        """reference :941-962"""
        if self.scheduler is None:
            return
        if hparams.iterations_per_scheduler_step:
            if current_iter is None or current_iter % hparams.iterations_per_scheduler_step != 0:
                return
        elif hparams.epochs_per_scheduler_step:
            if current_epoch is None or current_epoch % hparams.epochs_per_scheduler_step != 0:
                return
        else:
            raise ValueError("Scheduler {} is defined but neither hparams.iteration_per_scheduler"
                             "_step nor hparams.epochs_per_scheduler_step is set."
                             .format(hparams.scheduler_type))
        self._scheduler_step_fn(loss)

    @staticmethod
    def _get_current_iteration(batch_index, current_epoch, dataloader_length, hparams,
                               total_epoch):
        epoch = total_epoch if hparams.use_saved_learning_rate else current_epoch
        assert epoch is not None
        return (epoch - 1) * dataloader_length + batch_index + 1

    @staticmethod
    def get_summed_losses_subset(loss_names, losses):
        chosen = list(losses.values()) if loss_names is None else [losses[name] for name in loss_names]
        if len(chosen) == 1:
            return chosen[0]                    # (sum() would launch 0 + loss)
        return sum(chosen)

    # ------------------------------------------------------------------- train / validation
    def _to_device(self, data, device, non_blocking=True):
        return {k: (v.to(device, non_blocking=non_blocking) if torch.is_tensor(v) else v)
                for k, v in data.items()}

    def _device(self):
        return next(self.model.parameters()).device

    def test(self, hparams, total_epoch, total_steps, current_epoch):
        return self.process_dataloader(self.dataloader_val, hparams, total_epoch, total_steps,
                                       current_epoch, training=False)

    def train(self, hparams, total_epoch, total_steps, current_epoch):
        if hparams.ema_decay and not self.ema:
            self.ema = ExponentialMovingAverage(self.model, hparams.ema_decay)
            if isinstance(self.optimiser, HipAdam):
                self.optimiser.attach_ema(self.ema)       # updated inside the fused step
        return self.process_dataloader(self.dataloader_train, hparams, total_epoch, total_steps,
                                       current_epoch, training=True)

    def process_dataloader(self, dataloader, hparams, total_epoch, total_steps,
                           current_epoch=None, training=True):
        """One pass over the loader (reference :683-882): per mini-batch forward, named losses,
        NaN / Inf checks, and when training backward, optional clipping, optimiser, EMA and
        scheduler; returns {loss name: mean over the mini-batches} as numpy scalars.
        Multi-GPU: one process per GPU under torch.distributed (idiaptts_amd.parallel) instead of
        the reference's in-process DataParallel -- every rank draws the same global batch, keeps
        its own samples, and the frame-count-weighted gradients are summed over RCCL, which
        reproduces the single-GPU step on the whole batch."""
        if self._resident is not None and dataloader in (self.dataloader_train,
                                                            self.dataloader_val):
            reason = self._resident_unsupported(hparams, training)
            if reason is None:
                key = "train" if dataloader is self.dataloader_train else "val"
                return self._process_resident(dataloader, self._resident["shards"][key], hparams,
                                              total_epoch, total_steps, current_epoch, training)
            dataloader = self._resident_fall_through(hparams, reason, dataloader)
        model = self.model
        if training:
            model.train()
            self.logger.info("{}: Train with {} on {}.".format(
                datetime.now().strftime("%Y-%m-%d %H:%M:%S"), type(self.optimiser).__name__,
                self._device()))
        else:
            if self.ema is not None:
                self.logger.info("Using averaged model for validation.")
                model = self.ema.model
            model.eval()
        device = self._device()
        logging_batch_index = (len(dataloader) // hparams.logging_batch_index_perc) + 1
        total_losses = dict()
        # reference :694-705, :726-729: scalar writer and the memory line of a pass; the per-step NaN / Inf guard
        # (:778-781) is looked at one step late instead of stalling the host on every loss (logging_sinks)
        scalars = logging_sinks.DeferredScalars(logging_sinks.open_scalar_writer(hparams))
        guard = logging_sinks.DeferredLossCheck(device, check_inf=not hparams.replace_inf_grads_by_zero)
        if hparams.log_memory_consumption:
            self.logger.info(logging_sinks.memory_message(hparams.use_gpu))
        # clipping inside the optimiser's fused step when it can (HipAdam, norm types 2 / inf)
        fused_clip = False
        if training and hasattr(self.optimiser, "configure_clipping"):
            fused_clip = self.optimiser.configure_clipping(
                hparams.grad_clip_norm_type, hparams.grad_clip_max_norm, hparams.grad_clip_thresh)
        for batch_index, (data_dict, lengths) in enumerate(dataloader):
            data_dict = self._to_device(data_dict, device, hparams.dataset_load_async)
            batch_size = len(next(iter(lengths.values())))
            model.init_hidden(batch_size)
            # (the built-in max walks a tensor element by element through Python: 0.4 ms per key and batch)
            max_lengths = {k: (lengths[k].max() if torch.is_tensor(lengths[k]) else max(lengths[k]))
                           for k in data_dict if k in lengths}
            with torch.enable_grad() if training else torch.no_grad():
                with padding_rows_identical(self._stock_padding and dataloader in (self.dataloader_train,
                                                                                   self.dataloader_val)):
                    model(data_dict, lengths, max_lengths)
                losses = {}
                for loss_fn in self.losses:
                    for loss_name, l in loss_fn(data_dict, lengths, total_steps).items():
                        if loss_name in losses:
                            raise KeyError("Loss with name {} defined twice.".format(loss_name))
                        losses[loss_name] = l
                guard.submit(losses)
                backprop_loss = self.get_summed_losses_subset(hparams.backprop_loss_names, losses)
            if hparams.backprop_loss_names is None and hparams.scheduler_loss_names is None:
                scheduler_loss = backprop_loss.detach()
            else:
                scheduler_loss = self.get_summed_losses_subset(
                    hparams.scheduler_loss_names, losses).detach()
            dp_weight = self._dp_weight(lengths, device)
            if training:
                self.optimiser.zero_grad()
                if dp_weight is not None and hasattr(self.optimiser, "begin_overlapped_allreduce"):
                    self.optimiser.begin_overlapped_allreduce(dp_weight)     # bucket reductions beside backward
                backprop_loss.backward(gradient=unit_gradient(backprop_loss) if backprop_loss.dim() == 0 else None,
                                       retain_graph=hparams.backward_retain_graph)
                total_steps += 1
                if dp_weight is not None:
                    flat_sync = getattr(self.optimiser, "allreduce_grads_", None)
                    if flat_sync is None or not flat_sync(dp_weight):
                        parallel.allreduce_module_grads_(list(self.model.parameters()), dp_weight)
                if hparams.replace_inf_grads_by_zero:
                    self._replace_inf_grads_by_zero()
                if hparams.grad_clip_norm_type is not None and not fused_clip:
                    torch.nn.utils.clip_grad_norm_(self.model.parameters(),
                                                   hparams.grad_clip_max_norm,
                                                   hparams.grad_clip_norm_type)
                if hparams.grad_clip_thresh is not None and not fused_clip:
                    torch.nn.utils.clip_grad_value_(self.model.parameters(),
                                                    hparams.grad_clip_thresh)
                self.optimiser.step()
                if self.ema:
                    self.ema.update_params(model)
                current_iter = self._get_current_iteration(
                    batch_index=batch_index, current_epoch=current_epoch,
                    dataloader_length=len(dataloader), hparams=hparams, total_epoch=total_epoch)
                self.run_scheduler(hparams=hparams, loss=scheduler_loss,
                                   current_iter=current_iter)
            if batch_index % logging_batch_index == 0 and self.logger.isEnabledFor(logging.INFO):
                # (the line reads the losses back: a host wait for the whole queue -- only when somebody listens)
                self.logger.info("{} mini batch [{}/{}]\tLoss: {}{}".format(
                    "Train" if training else "Test", batch_index + 1, len(dataloader),
                    " ".join("{}: {:.3f}".format(k, float(l.detach())) for k, l in losses.items()),
                    "\t" + logging_sinks.memory_message(hparams.use_gpu)
                    if hparams.log_memory_consumption else ""))
            step_losses = {}
            for key, loss in losses.items():
                loss = loss.detach()
                if dp_weight is not None:        # loss of the global batch
                    loss = parallel.allreduce_flat_(loss * dp_weight)
                step_losses[key] = loss
                total_losses[key] = loss if key not in total_losses else total_losses[key] + loss
            if training:
                scalars.add_scalars("Train loss", step_losses, total_steps)     # reference :858-859
        guard.finish()
        total_losses = {k: v / len(dataloader) for k, v in total_losses.items()}
        if not training:
            scalars.add_scalars("Validation loss", total_losses, total_steps)   # reference :866-867
        scalars.flush()
        if not training:
            self.logger.info('Validation set: Total loss: {}\nAverage loss:\n\t{}\n'.format(
                float(sum(total_losses.values())),
                "\n\t".join("{}: {:.3f}".format(k, float(l)) for k, l in total_losses.items())))
            fn_log_per_test = getattr(self.model, "log_per_test", None)
            if callable(fn_log_per_test):
                fn_log_per_test()
        return {k: l.cpu().numpy() for k, l in total_losses.items()}

    def _dp_weight(self, lengths, device):
        """n_local / n_global valid frames of this step under data parallelism (None on a single
        process): the losses are means over the local frames (NamedLoss 'mean_per_frame'), so this
        weight turns the local loss and gradients into this rank's share of the global mean."""
        if parallel.dp_rank_world()[1] == 1:
            return None
        key = next((l.seq_mask for l in self.losses if getattr(l, "seq_mask", None) in lengths),
                   next(iter(lengths)))
        n_local = float(lengths[key].sum()) if torch.is_tensor(lengths[key]) else float(sum(lengths[key]))
        return n_local / parallel.global_sum(n_local, device=device)

    def _replace_inf_grads_by_zero(self):
        for p in self.model.parameters():
            if p.grad is not None:
                p.grad[torch.isinf(p.grad)] = 0.0

    def process_batch(self, data, lengths, step, training=True, grad_clip_norm=None, blocking=True):
        """One iteration of process_dataloader (:745-831) on a caller-supplied batch.  Returns
        ({loss name: float}, data).  With blocking=False nothing waits for the device: the losses come
        back as 0-dim device tensors and the NaN guard (reference :778-781) of this step is looked at when
        the next call has queued its work -- `finish_batches()` looks at the last one."""
        device = self._device()
        data = self._to_device(data, device)
        max_lengths = {k: int(v.max()) for k, v in lengths.items()}
        batch_dim = 0 if self.model.batch_first else 1
        B = data[self.model.input_names[0]].shape[batch_dim]
        self.model.init_hidden(B)
        with torch.enable_grad() if training else torch.no_grad():
            self.model(data, lengths, max_lengths)
            loss_dict = {}
            for loss_fn in self.losses:
                loss_dict.update(loss_fn(data, lengths, step))
            total = self.get_summed_losses_subset(None, loss_dict)
            if blocking:
                if torch.isnan(total):
                    raise ValueError("Found NaN in loss.")       # reference :778-781
            else:
                if self._batch_guard is None:
                    self._batch_guard = logging_sinks.DeferredLossCheck(device, check_inf=False,
                                                                       nan_message="Found NaN in loss.")
                self._batch_guard.submit({"summed": total})
            dp_weight = self._dp_weight(lengths, device)
            if training:
                self.optimiser.zero_grad()
                if dp_weight is not None and hasattr(self.optimiser, "begin_overlapped_allreduce"):
                    self.optimiser.begin_overlapped_allreduce(dp_weight)     # bucket reductions beside backward
                total.backward(gradient=unit_gradient(total) if total.dim() == 0 else None)
                if dp_weight is not None:
                    flat_sync = getattr(self.optimiser, "allreduce_grads_", None)
                    if flat_sync is None or not flat_sync(dp_weight):
                        parallel.allreduce_module_grads_(list(self.model.parameters()), dp_weight)
                if grad_clip_norm is not None:
                    torch.nn.utils.clip_grad_norm_(self.model.parameters(), grad_clip_norm)
                self.optimiser.step()
                if self.ema:
                    self.ema.update_params(self.model)
        if dp_weight is not None:
            loss_dict = {k: parallel.allreduce_flat_(v.detach() * dp_weight)
                         for k, v in loss_dict.items()}
        if not blocking:
            return {k: v.detach() for k, v in loss_dict.items()}, data
        return {k: float(v.detach()) for k, v in loss_dict.items()}, data

    def finish_batches(self):
        """The deferred NaN guard of the last process_batch(blocking=False) call."""
        if self._batch_guard is not None:
            self._batch_guard.finish()

    # ----------------------------------------------------------------------------- inference
    def inference(self, data, hparams, seq_lengths):
        """reference :964-993: eval mode, numpy / tensors in, numpy out (keys starting with '_'
        are dropped)."""
        self._resident_sync_to_module()
        self.model.eval()
        to_torch = lambda v: torch.from_numpy(v) if isinstance(v, np.ndarray) else v  # noqa: E731
        device = self._device()
        data_torch = self._to_device({k: to_torch(v) for k, v in data.items()}, device)
        lengths_torch = {k: to_torch(v) for k, v in seq_lengths.items()}
        max_lengths = {k: max(v) for k, v in seq_lengths.items()}
        self.model.init_hidden(len(next(iter(seq_lengths.values()))))
        with torch.no_grad():
            self.model.inference(data_torch, lengths_torch, max_lengths)
        out = {k: self._return_values_to_numpy(v) for k, v in data_torch.items()
               if not k.startswith('_')}
        out_lengths = {k: self._return_values_to_numpy(v) for k, v in lengths_torch.items()
                       if not k.startswith('_')}
        return out, out_lengths

    @staticmethod
    def _return_values_to_numpy(value, from_gpu=None):
        if value is None:
            return None
        if isinstance(value, (tuple, list)):
            return tuple(ModularModelHandlerPyTorch._return_values_to_numpy(v) for v in value)
        if torch.is_tensor(value):
            return value.detach().cpu().numpy()
        return value

    # -------------------------------------------------------------------------- checkpoints
    def save_checkpoint(self, model_path, best_loss=np.inf, epoch=None, step=None,
                        save_as_best_model=False, save_as_epoch=True, save_as_last_model=False,
                        save_as_step=True):
        """reference :71-123"""
        assert save_as_best_model or save_as_last_model or step is not None \
            or epoch is not None, "Epoch or step needs to be given."
        assert model_path is not None, "Given model_path cannot be None."
        if save_as_best_model:
            suffix = "best"
        elif save_as_last_model:
            suffix = "last"
        elif epoch is not None and save_as_epoch:
            suffix = "e{}".format(epoch)
        elif step is not None and save_as_step:
            suffix = "s{}".format(step)
        else:
            raise NotImplementedError()
        self._resident_sync_to_module()
        if parallel.dp_rank_world()[0] != 0:
            # data parallel: the replicas are identical, rank 0 writes; nobody runs ahead of the
            # files (a load may follow)
            parallel.barrier()
            return
        self.logger.info("Save {} checkpoint to {}.".format(suffix, model_path))
        os.makedirs(model_path, exist_ok=True)
        config = self.model_config if self.model_config is not None \
            else getattr(self.model, "config", None)
        if config is not None:
            with open(os.path.join(model_path, "config.json"), "w") as f:
                f.write(config_json.encode(config))
        params = self.model.state_dict()
        if self.ema:
            params.update(self.ema.shadow)      # only the shadowed (trainable) parameters
        if getattr(self, "async_checkpoint", False):
            cls = type(self)
            if cls._checkpoint_writer is None:
                cls._checkpoint_writer = _CheckpointWriter()
            save = cls._checkpoint_writer.submit
        else:
            save = torch.save
        save({"params": params, "epoch": epoch, "step": step},
             os.path.join(model_path, "params_" + suffix))
        if self.optimiser is not None:
            save({"params": self.optimiser.state_dict(), "epoch": epoch, "step": step,
                  "best_loss": best_loss}, os.path.join(model_path, "optimiser_" + suffix))
        if self.scheduler is not None:
            save({"params": self.scheduler.state_dict(), "epoch": epoch, "step": step},
                 os.path.join(model_path, "scheduler_" + suffix))
        if parallel.dp_rank_world()[1] > 1:
            self.wait_for_checkpoints()
            parallel.barrier()

    def load_checkpoint(self, hparams, model_path, epoch=None, ignore_layers=True,
                        load_optimiser=True, load_scheduler=True, step=None, verbose=True,
                        load_best_model=False):
        """reference :125-262.  Returns (best_loss, epoch, step)."""
        self.wait_for_checkpoints()
        parallel.barrier()
        assert load_best_model or step is None or epoch is None, \
            "Only epoch ({}) OR step ({}) can be not None".format(epoch, step)
        if load_best_model or epoch == -1 or step == -1:
            suffix = "_best"
        elif hparams.load_newest_checkpoint:
            assert step is None and epoch is None
            file_list = glob.glob(os.path.join(model_path, "params_*"))
            if len(file_list) == 0:
                raise FileNotFoundError("No newest checkpoint found in {}.".format(model_path))
            if len(file_list) > 1:
                file_list = [f for f in file_list
                             if os.path.basename(f) not in ["params_e0", "params_s0"]]
            suffix = "_" + os.path.basename(max(file_list, key=os.path.getctime)).split('_')[1]
        else:
            assert step is not None or epoch is not None, \
                "Either step or epoch is required. Use -1 in one of them to load the best model."
            suffix = "_s{}".format(step) if step is not None else "_e{}".format(epoch)
        params_path = os.path.join(model_path, "params" + suffix)
        world = parallel.dp_rank_world()[1]
        if world > 1:
            # every rank loads the files (the optimiser and scheduler state come from them) and rank 0's tensors are
            # broadcast behind that: a rank that cannot see a file must not be the only one to raise -- the others
            # would wait for it in the broadcast for ever (ranks with out_dirs of their own, a file system not shared)
            seen = parallel.global_sum(1.0 if os.path.isfile(params_path) else 0.0, device=self._dist_device())
            if seen < world:
                raise FileNotFoundError("{} is visible to {} of the {} ranks (they need one shared out_dir)."
                                        .format(params_path, int(seen), world))
        if verbose:
            self.logger.info("Load model state dict from {}".format(params_path))
        checkpoint = torch.load(params_path, map_location="cpu", weights_only=False)
        params = checkpoint["params"] if "params" in checkpoint \
            else checkpoint["model_state_dict"]
        best_loss = np.inf
        epoch = checkpoint["epoch"]
        step = checkpoint.get("step")
        if self.model is None:
            with open(os.path.join(model_path, "config.json"), "r") as f:
                self.model_config = config_json.decode(f.read())
            self.model = self.model_config.create_model()
        if hparams.has_value("layer_map") and len(hparams.layer_map) > 0:
            params = self._map_layer_names(params, hparams.layer_map, verbose)
        if ignore_layers:
            params = self._remove_ignored_layers(params, self.model, hparams)
        missing_keys, unexpected_keys = self.model.load_state_dict(
            params, strict=not hparams.allow_missing_layers)
        if verbose and len(missing_keys) > 0:
            self.logger.warning("Did not load: {}".format(", ".join(missing_keys)))
        if verbose and len(unexpected_keys) > 0:
            self.logger.warning("Found unexpected keys: {}".format(", ".join(unexpected_keys)))
        if hparams.use_gpu:
            self.model = self.model.cuda()
        if load_optimiser:
            checkpoint = torch.load(os.path.join(model_path, "optimiser" + suffix),
                                    map_location="cpu", weights_only=False)
            if "best_loss" in checkpoint and (not ignore_layers or not hparams.ignore_layers):
                best_loss = checkpoint["best_loss"]
            self._load_optimiser(checkpoint["params"], hparams)
            scheduler_path = os.path.join(model_path, "scheduler" + suffix)
            if load_scheduler and os.path.isfile(scheduler_path):
                sched = torch.load(scheduler_path, map_location="cpu", weights_only=False)
                self._load_scheduler(sched["params"],
                                     epoch if epoch is not None else sched['epoch'],
                                     step if step is not None else sched['step'], hparams)
        if self._resident is not None:
            self._resident["synced"] = False      # flat buffers follow the loaded state
        self._reseed_ema()
        self.sync_from_rank0()
        return best_loss, epoch, step

    @staticmethod
    def _map_layer_names(params, layer_map, verbose=True):
        new_params = {}
        for name, param in params.items():
            for pattern, replacement in layer_map:
                if re.search(pattern, name) is not None:
                    name = re.sub(pattern, replacement, name)
                    break
            new_params[name] = param
        return new_params

    @staticmethod
    def _remove_ignored_layers(model_dict, model, hparams):
        """Layers matching hparams.ignore_layers keep the model's current values (:285-309)."""
        ignore = getattr(hparams, "ignore_layers", None) or []
        keys_to_pop = []
        for ignored_layer in ignore:
            found = [k for k in model_dict if re.match(ignored_layer, k)]
            if not found:
                raise KeyError("Cannot find layer {} in saved model dict: {}".format(
                    ignored_layer, ", ".join(model_dict.keys())))
            keys_to_pop += found
        if keys_to_pop:
            org = model.state_dict()
            model_dict = {k: v for k, v in model_dict.items() if k not in keys_to_pop}
            model_dict.update({k: org[k] for k in keys_to_pop if k in org})
        return model_dict

    def _load_optimiser(self, opt_params, hparams):
        self.set_optimiser(hparams, reset=True)
        try:
            self.optimiser.load_state_dict(opt_params)
        except ValueError as e:
            self.logger.warning("Optimiser state of the checkpoint does not match {}: {}\n"
                                "Continuing without loading it.".format(
                                    hparams.optimiser_type, e))
        # torch's load_state_dict casts state tensors to the parameter's device already.
        if "lr" in hparams.optimiser_args:
            for group in self.optimiser.param_groups:
                group.setdefault('initial_lr', hparams.optimiser_args["lr"])

    def _load_scheduler(self, scheduler_params, current_epoch, current_step, hparams):
        if hparams.epochs_per_scheduler_step is not None:
            self.set_scheduler(hparams, current_epoch=current_epoch, reset=True)
        elif hparams.iterations_per_scheduler_step is not None:
            self.set_scheduler(hparams, current_step=current_step, reset=True)
        else:
            self.set_scheduler(hparams, current_epoch=current_epoch, current_step=current_step,
                               reset=True)
        if self.scheduler is not None:
            try:
                self.scheduler.load_state_dict(scheduler_params)
            except (ValueError, KeyError) as e:
                self.logger.warning("Scheduler state of the checkpoint does not match {}: {}"
                                    .format(hparams.scheduler_type, e))
