"""What the reference logs around the training loop, without a host synchronisation per step:

* `get_gpu_memory_map()` -- the reference asks `nvidia-smi --query-gpu=memory.used`
  (idiaptts/misc/utils.py:152-175); here `torch.cuda.mem_get_info` per visible device (MB in use,
  the same {device index: MB} dictionary).
* `open_scalar_writer(hparams)` -- the reference opens a `torch.utils.tensorboard.SummaryWriter`
  in `<out_dir>/<model_name>/tensorboard` (or `hparams.tensorboard_dir`) and carries on without one
  when tensorboard is not installed (ModularModelHandlerPyTorch.py:694-705,
  model_trainers/ModularTrainer.py:198-214).  Same lookup here; when tensorboard is missing AND
  `hparams.scalar_log_fallback` is set (default on), scalars go to `<dir>/scalars.jsonl` instead,
  one JSON object per `add_scalars` call.
* `DeferredScalars` -- `add_scalars(tag, {name: 0-dim device tensor}, step)` as the reference calls it
  per mini-batch (:858-867), but the values stay on the device until `flush()` (every
  `flush_every` calls and at the end of a pass): one stack + one copy instead of a blocking
  `float()` per loss and step.
* `DeferredLossCheck` -- the reference's NaN / Inf guard (:778-781) raises before backward, which costs
  a host synchronisation per loss and step.  Here the finiteness flags of a step are computed on the
  device, copied to page-locked memory behind the step's kernels and looked at when the NEXT step
  has been queued (and at the end of the pass): the same ValueError with the same message, one
  step late."""
import json
import os
import resource

import torch


def get_gpu_memory_map():
    """{device index: MB in use} over the visible devices, "not available" without a GPU."""
    if not torch.cuda.is_available():
        return "not available"
    usage = {}
    for i in range(torch.cuda.device_count()):
        free, total = torch.cuda.mem_get_info(i)
        usage[i] = int((total - free) // (1024 * 1024))
    return usage


def memory_message(use_gpu):
    """'CPU: <MB> MB, GPU: {..} MB' as the reference formats it (handler :726-729)."""
    return "CPU: {:.0f} MB, GPU: {} MB".format(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e3,
                                                str(get_gpu_memory_map()) if use_gpu else "-")


class JsonlScalarWriter(object):
    """SummaryWriter's add_scalars / add_text / flush / close on a JSON-lines file."""

    def __init__(self, log_dir):
        os.makedirs(log_dir, exist_ok=True)
        self.path = os.path.join(log_dir, "scalars.jsonl")
        self._f = open(self.path, "a")

    def add_scalars(self, main_tag, tag_scalar_dict, global_step=None):
        self._f.write(json.dumps({"tag": main_tag, "step": None if global_step is None else int(global_step),
                                  "scalars": {k: float(v) for k, v in tag_scalar_dict.items()}}) + "\n")

    def add_text(self, tag, text_string, global_step=None):
        self._f.write(json.dumps({"tag": tag, "step": global_step, "text": text_string}) + "\n")

    def flush(self):
        self._f.flush()

    def close(self):
        self._f.close()


_writers = {}


def open_scalar_writer(hparams):
    """One writer per log directory and process (the reference opens one in the trainer and one per
    process_dataloader call; tensorboard tolerates that, a shared file does not need it)."""
    if hparams.has_value("tensorboard_dir"):
        log_dir = hparams.tensorboard_dir
    elif hparams.out_dir is not None and hparams.has_value("model_name"):
        log_dir = os.path.join(hparams.out_dir, hparams.model_name, "tensorboard")
    else:
        return None
    log_dir = os.path.abspath(log_dir)
    if log_dir in _writers:
        return _writers[log_dir]
    try:
        from torch.utils.tensorboard import SummaryWriter
        writer = SummaryWriter(log_dir=log_dir)
    except ImportError:
        fallback = hparams.scalar_log_fallback if hparams.has_value("scalar_log_fallback") else True
        writer = JsonlScalarWriter(log_dir) if fallback else None
    _writers[log_dir] = writer
    return writer


class DeferredScalars(object):
    def __init__(self, writer, flush_every=64):
        self.writer = writer
        self.flush_every = flush_every
        self._pending = []        # (tag, names, stacked device tensor or list of tensors, step)

    def add_scalars(self, tag, scalars, step):
        if self.writer is None or not scalars:
            return
        names = list(scalars)
        self._pending.append((tag, names, torch.stack([scalars[n].detach().reshape(()).float() for n in names]), step))
        if len(self._pending) >= self.flush_every:
            self.flush()

    def flush(self):
        if self.writer is None or not self._pending:
            return
        values = torch.cat([p[2] for p in self._pending]).cpu().tolist()      # one copy for all of them
        at = 0
        for tag, names, _, step in self._pending:
            self.writer.add_scalars(tag, dict(zip(names, values[at:at + len(names)])), step)
            at += len(names)
        self._pending = []
        self.writer.flush()


class DeferredLossCheck(object):
    def __init__(self, device, check_inf=True, nan_message=None):
        self.check_inf = check_inf
        self.nan_message = nan_message      # instead of "Found NaN in <name> loss."
        self.on_device = torch.device(device).type == "cuda"
        self._slots = []
        if self.on_device:
            for _ in range(2):
                self._slots.append({"host": torch.zeros(64, dtype=torch.int32).pin_memory(),
                                    "event": torch.cuda.Event(), "names": None})
        self._turn = 0

    def _raise(self, names, codes):
        for name, code in zip(names, codes):
            if code & 1:
                raise ValueError(self.nan_message or "Found NaN in {} loss.".format(name))
            if (code & 2) and self.check_inf:
                raise ValueError("Found +/-Inf in {} loss.".format(name))

    def submit(self, losses):
        """losses: {name: 0-dim tensor} of the step just queued.  Looks at the step before."""
        names = list(losses)
        if not self.on_device:
            vals = torch.stack([losses[n].detach().reshape(()) for n in names])
            self._raise(names, (torch.isnan(vals).int() + 2 * torch.isinf(vals).int()).tolist())
            return
        assert len(names) <= 64
        slot = self._slots[self._turn]
        other = self._slots[1 - self._turn]
        vals = torch.stack([losses[n].detach().reshape(()) for n in names])
        codes = torch.isnan(vals).to(torch.int32) + 2 * torch.isinf(vals).to(torch.int32)
        slot["host"][:len(names)].copy_(codes, non_blocking=True)
        slot["event"].record()
        slot["names"] = names
        self._turn = 1 - self._turn
        self._look(other)

    def _look(self, slot):
        if slot["names"] is None:
            return
        slot["event"].synchronize()       # the step before this one: long done unless the host runs ahead
        names, slot["names"] = slot["names"], None
        self._raise(names, slot["host"][:len(names)].tolist())

    def finish(self):
        for slot in self._slots:
            self._look(slot)
