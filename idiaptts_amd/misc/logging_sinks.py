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
  a host synchronisation per loss and step.  Here the loss values of a step are copied to
  page-locked memory behind the step's kernels (one small copy per loss, nothing else on the device)
  and classified when the NEXT step has been queued (and at the end of the pass): the same
  ValueError with the same message, one step late."""
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
                self._slots.append({"host": torch.zeros(64, dtype=torch.float32).pin_memory(),
                                    "host64": torch.zeros(64, dtype=torch.float64).pin_memory(),
                                    "event": torch.cuda.Event(), "names": None, "wide": None})
        self._turn = 0

    def _raise(self, names, values):
        for name, value in zip(names, values):
            if value != value:
                raise ValueError(self.nan_message or "Found NaN in {} loss.".format(name))
            if value in (float("inf"), float("-inf")) and self.check_inf:
                raise ValueError("Found +/-Inf in {} loss.".format(name))

    def submit(self, losses):
        """losses: {name: 0-dim tensor} of the step just queued.  Looks at the step before."""
        names = list(losses)
        if not self.on_device:
            self._raise(names, [float(losses[n].detach()) for n in names])
            return
        assert len(names) <= 64
        slot = self._slots[self._turn]
        other = self._slots[1 - self._turn]
        # the values themselves go to page-locked memory, one small copy per loss, and are classified on the host a
        # step later (until round 6 the device classified them: isnan, isinf, two casts, a product, a sum -- nine
        # launches of 5 us each on a step that is 1.3 ms of device time)
        wide = []
        for i, n in enumerate(names):
            v = losses[n].detach().reshape(1)
            wide.append(v.dtype == torch.float64)
            if v.dtype not in (torch.float32, torch.float64):
                v = v.float()
            slot["host64" if wide[-1] else "host"][i:i + 1].copy_(v, non_blocking=True)
        slot["event"].record()
        slot["names"], slot["wide"] = names, wide
        self._turn = 1 - self._turn
        self._look(other)

    def _look(self, slot):
        if slot["names"] is None:
            return
        slot["event"].synchronize()       # the step before this one: long done unless the host runs ahead
        names, slot["names"] = slot["names"], None
        narrow, wide = slot["host"][:len(names)].tolist(), slot["host64"][:len(names)].tolist()
        self._raise(names, [w if is_wide else v for v, w, is_wide in zip(narrow, wide, slot["wide"])])

    def finish(self):
        for slot in self._slots:
            self._look(slot)
